"""Import-path compatibility with the reference's notebooks/infer_s2.py (stage-2 wrapper)."""
from emoportraits_amd.stage2 import InferenceWrapper  # noqa: F401
