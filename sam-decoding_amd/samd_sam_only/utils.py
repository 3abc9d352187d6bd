"""Names the reference keeps in samd_sam_only/utils.py, gathered from where they live here."""
from .draft import Candidates, CandidateType, DraftModel  # noqa: F401
from .generation import SamdGenerationConfig  # noqa: F401
from .posterior import OptionalTensor, device_argmax, eval_posterior, eval_posterior_nodes, gen_candidates  # noqa: F401
