"""Import-path compatibility with the reference: `from notebooks.infer import InferenceWrapper` (or `from infer import
InferenceWrapper` with notebooks/ on sys.path, as the reference's notebooks do) resolves to the MI355X implementation."""
from emoportraits_amd.infer import InferenceWrapper, HipModel  # noqa: F401
