from .dpo import DPOTrainer  # noqa: F401
