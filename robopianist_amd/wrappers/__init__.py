from robopianist_amd.wrappers.canonical import CanonicalSpecWrapper
from robopianist_amd.wrappers.evaluation import MidiEvaluationWrapper

__all__ = ["CanonicalSpecWrapper", "MidiEvaluationWrapper"]
