"""samd/utils.py of the reference is byte-identical to samd_sam_only/utils.py; one
implementation serves both packages."""
from samd_sam_only.utils import (OptionalTensor, SamdGenerationConfig, device_argmax, eval_posterior,  # noqa: F401
                                 gen_candidates)
from .draft import Candidates, CandidateType, DraftModel  # noqa: F401
