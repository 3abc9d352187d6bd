"""samd/cache.py of the reference equals samd_sam_only/cache.py modulo imports; one implementation serves both."""
from samd_sam_only.cache import SamdCache, SamdStaticCache  # noqa: F401
