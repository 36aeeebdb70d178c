from .hip_graph_runner import HipGraphRunner
from .hip_vec_runner import EpisodeRunner, HipVecRunner

REGISTRY = {"episode": EpisodeRunner, "hip_vec": HipVecRunner, "hip_graph": HipGraphRunner}
