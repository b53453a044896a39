"""Shared body of the full-WIDTH parity tests against the reference trainer's fixtures (tests/golden/*_width_dpo.npz, oracle/gen_golden.py::gen_*_width):
the native DPO pair -- fp32 twin and bf16 production path -- against the reference's fp32 run, with the bf16 path held to the envelope DERIVED from the
reference's OWN bf16 run of the same fixture (native-bf16-vs-fp32 <= 1.5 x reference-bf16-vs-fp32, per quantity)."""
import gc

import numpy as np
import torch

from tests.gpu_util import dev, dump
from tests.util import rel_err


def width_parity(z, hf_config, sd, ref_sd, batch, pad_token_id, report, *, extra_train_cfgs=None, trainer_kwargs=None, batch_keys=(), float_keys=(),
                 skip_norm_of=(), min_matrices=29, fp32_bounds=(2e-4, 2e-4, 1e-3, 2e-3), check_vectors=False, expect_packed_rows=None):
    """z: the fixture; (hf_config, sd, ref_sd, batch): the regenerated model / pair (oracle.synthetic.*_width); batch_keys: extra batch entries handed to the
    trainer as they are (grids, masks); float_keys: batch entries cast to the compute dtype (pixels, mel features); skip_norm_of: parameters whose stored
    layout differs from HF's (norm compared through the others).  fp32_bounds: loss abs, per-token log-probs abs, gradient-norm rel, leading-block rel_err.
    check_vectors: the 1-D gradients the fixture holds (norm weights, biases) are compared as well (norm rel, fp32 asserted at fp32_bounds[2])."""
    from align_anything_amd import configs
    from align_anything_amd.trainers.dpo import DPOTrainer
    T = torch.from_numpy
    names = [str(n) for n in z['names']]
    for n, c, rc in zip(names, z['weight_checksum'], z['ref_weight_checksum']):       # the identical weights were regenerated from the seed
        assert abs(float(sd[n].double().sum()) - float(c)) <= 1e-9 * max(1.0, abs(float(c))), n
        assert abs(float(ref_sd[n].double().sum()) - float(rc)) <= 1e-9 * max(1.0, abs(float(rc))), n
    assert np.array_equal(batch['input_ids'].numpy(), z['input_ids'])
    cfg = configs.from_hf_config(hf_config)
    want_lp, want_ref = T(z['seq_log_probs']), T(z['ref_seq_log_probs'])
    rep = [f'reference trainer (fp32, CPU): loss {float(z["loss_loss"]):.6f} margin {z["loss_reward_margin"].tolist()} summed log-probs {want_lp.sum(1).tolist()}']
    try:
        for dtype in ('fp32', 'bf16'):
            tc = {'scale_coeff': float(z['scale_coeff']), 'learning_rate': 1e-6, 'lr_warmup_ratio': 0.0, 'lr_scheduler_type': 'constant', 'compute_dtype': dtype}
            tc.update(extra_train_cfgs or {})
            tr = DPOTrainer({'train_cfgs': tc, 'model_cfgs': {'pad_token_id': int(pad_token_id)}}, {'gradient_clipping': 1.0}, model_cfg=cfg, policy_state=sd,
                            reference_state=ref_sd, device='cuda:0', **(trainer_kwargs or {}))
            b = {'input_ids': batch['input_ids'].to(dev()), 'attention_mask': batch['attention_mask'].to(dev()), 'meta_info': batch['meta_info']}
            for k in batch_keys:
                b[k] = batch[k]
            for k in float_keys:
                b[k] = batch[k].to(dev()).to(torch.float32 if dtype == 'fp32' else torch.bfloat16)
            lp = tr.compute_log_probs(tr.model, b).cpu()
            if expect_packed_rows is not None:            # the shared-prompt layout was really taken
                assert b.get('_pack') is not None and b['_pack']['rows'] == expect_packed_rows, (b.get('_pack') or {}).get('rows')
                rep.append(f'shared-prompt packing: {b["_pack"]["rows"]} token rows instead of {int(batch["attention_mask"].sum())} ({b["_pack"]["shared_rows"]} shared)')
            rlp = tr.compute_log_probs(tr.reference_model, b).cpu()
            assert torch.equal(lp == 0, want_lp == 0), 'response-window layout differs from the reference'
            ld = tr.loss(b)
            tr.model.backward(ld['loss'])
            torch.cuda.synchronize()
            m = {'loss': abs(float(ld['loss']) - float(z['loss_loss'])),
                 'margin': float((ld['reward_margin'].float().cpu().reshape(-1) - T(z['loss_reward_margin']).reshape(-1)).abs().max()),
                 'per-token log-probs (policy)': float((lp - want_lp).abs().max()), 'per-token log-probs (reference model)': float((rlp - want_ref).abs().max()),
                 'summed log-probs': float((lp.sum(1) - want_lp.sum(1)).abs().max())}
            r = {'loss': abs(float(z['bf16.loss_loss']) - float(z['loss_loss'])),
                 'margin': float(np.abs(z['bf16.loss_reward_margin'].reshape(-1) - z['loss_reward_margin'].reshape(-1)).max()),
                 'per-token log-probs (policy)': float(np.abs(z['bf16.seq_log_probs'] - z['seq_log_probs']).max()),
                 'per-token log-probs (reference model)': float(np.abs(z['bf16.ref_seq_log_probs'] - z['ref_seq_log_probs']).max()),
                 'summed log-probs': float(np.abs(z['bf16.seq_log_probs'].sum(1) - z['seq_log_probs'].sum(1)).max())}
            # The trainer's loss is formed from FOUR fp32 sums of ~R log-probs each (reference: `.sum(-1)` on fp32 tensors, dpo.py:129-141); at |sum| ~ 5.7e3
            # one fp32 ulp of a sum is 4.9e-4, i.e. 4.9e-5 on beta x margin -- the summation's own rounding, in the reference as much as here.  So the
            # fp32 loss / margin are held to 4 x beta x ulp(max |sum|) (or the stated bound, whichever is larger), and the kernels' OWN contribution is
            # isolated by re-forming the loss in fp64 from the per-token log-probs of both sides (no summation rounding on either): held to the stated bound.
            beta = float(z['scale_coeff'])
            ulp = float(np.spacing(np.float32(max(float(want_lp.sum(1).abs().max()), float(want_ref.sum(1).abs().max())))))

            def loss64(p_lp, r_lp):
                d = (p_lp.double().sum(1) - r_lp.double().sum(1))
                h = d.shape[0] // 2
                return torch.nn.functional.softplus(-beta * (d[:h] - d[h:])).mean()
            m['loss re-formed in fp64 from the per-token log-probs'] = abs(float(loss64(lp, rlp)) - float(loss64(want_lp, want_ref)))
            r['loss re-formed in fp64 from the per-token log-probs'] = abs(float(loss64(T(z['bf16.seq_log_probs']), T(z['bf16.ref_seq_log_probs']))) - float(loss64(want_lp, want_ref)))
            wn, wb, rn, rb, n_g, wv = 0.0, 0.0, 0.0, 0.0, 0, 0.0
            for n, gn, gnb in zip(names, z['grad_norm'], z['bf16.grad_norm']):
                if gn <= 0:
                    continue
                g = tr.policy.store.grad_view(n)
                assert g is not None, n
                if len(g.shape) < 2:
                    if check_vectors:
                        ev = abs(float(g.float().double().norm()) - float(gn)) / float(gn)
                        wv = max(wv, ev)
                        assert dtype != 'fp32' or ev < fp32_bounds[2], (n, ev)
                    continue
                gf = g.float()
                n_g += 1
                if n in skip_norm_of:
                    continue
                wn = max(wn, abs(float(gf.double().norm()) - float(gn)) / float(gn))
                rn = max(rn, abs(float(gnb) - float(gn)) / float(gn))
                if 'gblk.' + n in z.files:
                    blk = T(z['gblk.' + n])
                    if float(blk.norm()) > 1e-3 * float(gn) / max(1.0, (gf.numel() / blk.numel()) ** 0.5):
                        wb = max(wb, rel_err(gf.reshape(gf.shape[0], -1)[:32, :32].cpu(), blk))
                        rb = max(rb, rel_err(T(z['bf16.gblk.' + n]), blk))
            m['worst matrix gradient norm (rel)'], r['worst matrix gradient norm (rel)'] = wn, rn
            m['worst leading gradient block (rel_err)'], r['worst leading gradient block (rel_err)'] = wb, rb
            rep.append(f'{dtype}: loss {float(ld["loss"]):.6f}; ' + '; '.join(f'{k} {v:.2e}' for k, v in m.items()) + f' ({n_g} matrices)'
                       + (f'; worst vector gradient norm (rel) {wv:.2e}' if check_vectors else ''))
            if dtype == 'fp32':
                sum_bound = max(fp32_bounds[0], 4 * beta * ulp)
                rep.append(f'  fp32 resolution of the summed log-probs: ulp {ulp:.2e} -> loss / margin bound {sum_bound:.2e}; kernels alone (fp64 re-formed loss) bound {fp32_bounds[0]:.1e}')
                assert m['loss'] < sum_bound and m['margin'] < sum_bound and m['loss re-formed in fp64 from the per-token log-probs'] < fp32_bounds[0] \
                    and m['per-token log-probs (policy)'] < fp32_bounds[1] and m['per-token log-probs (reference model)'] < fp32_bounds[1] \
                    and wn < fp32_bounds[2] and wb < fp32_bounds[3], rep[-1]
            else:
                rep.append('bf16 envelope, native vs the reference\'s own bf16 run (both against the reference\'s fp32 run):')
                for k in m:
                    rep.append(f'  {k}: native {m[k]:.3e}   reference bf16 {r[k]:.3e}   ratio {m[k] / max(r[k], 1e-30):.2f}')
                for k in m:
                    assert m[k] <= 1.5 * r[k], (k, m[k], r[k], rep)
            assert n_g >= min_matrices, n_g
            del tr
            gc.collect()
            torch.cuda.empty_cache()
    finally:
        dump(report, '\n'.join(rep) + '\n')
