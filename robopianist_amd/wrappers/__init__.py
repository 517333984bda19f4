from robopianist_amd.wrappers.canonical import CanonicalSpecWrapper
from robopianist_amd.wrappers.evaluation import MidiEvaluationWrapper
from robopianist_amd.wrappers.graphed import GraphedStepWrapper

__all__ = ["CanonicalSpecWrapper", "MidiEvaluationWrapper", "GraphedStepWrapper"]
