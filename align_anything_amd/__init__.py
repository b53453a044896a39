"""MI355X-native DPO/PPO inner loop for PKU-Alignment/align-anything (drop-in for the reference's
trainers' hot path).  See DESIGN.md / INTEGRATION.md."""
__version__ = '0.1.0'
