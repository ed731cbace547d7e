"""MI355X-native batched UR5 grasp-rollout simulator (drop-in for the reference's GraspEnv /
MJ_Controller hot path). Heavy imports are lazy so that the model compiler and
the CPU-side tests work without torch or a GPU."""
from .model import CompiledModel, load_model  # noqa: F401

__version__ = "0.1.0"
