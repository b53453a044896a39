from .dpo import DPOTrainer  # noqa: F401
from .pref import KTOTrainer, ORPOTrainer, SimPOTrainer  # noqa: F401
