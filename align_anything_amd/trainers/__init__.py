from .dpo import DPOTrainer  # noqa: F401
from .pref import KTOTrainer, ORPOTrainer, SimPOTrainer  # noqa: F401
from .sft import SupervisedTrainer  # noqa: F401
