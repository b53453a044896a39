"""GPU: PPO's PTX mix-in (trainers/ppo.py::PPOTrainer.ptx_step = align_anything/trainers/text_to_text/ppo.py:400-408): a supervised
step on the actor with ptx_coeff * causal-LM loss.  Written after the round's GPU budget was spent -- first run is the driver's; kept
in its own, alphabetically last file."""
import pytest
import torch

from oracle import models as om
from oracle import rl_math as orl
from tests.gpu_util import dev
from tests.test_ppo_gpu import _setup

pytestmark = pytest.mark.gpu


def test_ptx_step_is_a_scaled_supervised_step_on_the_actor():
    tr, z, cfg, actor_sd, old_sd, rm_sd, ids, mask, start = _setup()
    assert tr.ptx_coeff == 16.0
    labels = ids.clone()
    labels[mask == 0] = -100
    labels[:, :start] = -100
    b = {'input_ids': ids.to(dev()), 'attention_mask': mask.to(dev()), 'labels': labels}
    before = {g: t.clone() for g, t in tr.actor_model.module.store.master.items()}
    info = tr.ptx_step(b)
    a = {k: v.float() for k, v in actor_sd.items() if k != 'lm_head.weight'}
    want = float(orl.sft_loss(om.opt_logits(a, cfg, ids, mask), labels))
    assert abs(info['train/ptx_loss'] - want) < 3e-2, (info['train/ptx_loss'], want)      # bf16 actor vs fp32 oracle; logged UNSCALED
    tr.actor_model.wait_optimizer()
    torch.cuda.synchronize()
    moved = sum(float((tr.actor_model.module.store.master[g] - before[g]).abs().sum()) for g in before)
    assert moved > 0.0 and tr.actor_model.global_steps == 1
