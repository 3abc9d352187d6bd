"""SamdGenerationConfig -- the per-call generation settings (reference surface: samd_sam_only/utils.py:30-63).

Field names and defaults are the reference's, because callers construct it by keyword
(evaluation/inference_sam_only.py:17-22).  Non-greedy decoding needs HuggingFace's logits warpers, which are imported
lazily so that the greedy hot path has no transformers dependency."""
from dataclasses import dataclass
from typing import Any, Optional


@dataclass
class SamdGenerationConfig:
    max_steps: int = 512
    max_new_tokens: int = 512
    max_cache_len: int = 2048
    greedy: bool = True
    temperature: float = 0.0
    top_p: float = 0.0
    top_k: int = 0
    logits_processor: Optional[Any] = None

    def __post_init__(self):
        if self.greedy:
            return
        assert self.temperature >= 1e-5
        self.logits_processor = self.prepare_logits_processor(self.temperature, self.top_p, self.top_k)

    @staticmethod
    def prepare_logits_processor(temperature: float = 0.0, top_p: float = 0.0, top_k: int = 0):
        """warpers in the reference's order -- temperature, nucleus, top-k -- each only when its setting is active."""
        from transformers.generation import logits_process as lp
        chain = lp.LogitsProcessorList()
        wanted = ((temperature >= 1e-5 and temperature != 1.0, lp.TemperatureLogitsWarper, temperature),
                  (1e-8 <= top_p < 1.0, lp.TopPLogitsWarper, top_p),
                  (top_k > 0, lp.TopKLogitsWarper, top_k))
        for active, warper, value in wanted:
            if active:
                chain.append(warper(value))
        return chain
