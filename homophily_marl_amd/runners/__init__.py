from .hip_vec_runner import EpisodeRunner, HipVecRunner

REGISTRY = {"episode": EpisodeRunner, "hip_vec": HipVecRunner}
