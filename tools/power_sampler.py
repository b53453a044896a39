"""Package power and shader clock of one GPU, sampled at >= 10 Hz into a file: the sidecar bench.py starts around its timed region
("power-limited" must be a measured field of the bench line, not a reading of smi beside it).

    python tools/power_sampler.py --device 0 --out /tmp/x.jsonl [--hz 20]        # runs until SIGTERM / stdin closes

A separate PROCESS, not a thread: the step's host thread is busy enqueueing launches and must not share an interpreter lock with a poller.
One JSON line per sample: {"t": time.time(), "power_w": .., "sclk_mhz": .., "src": ..}.  Sources, first that answers (recorded per line):
  amdsmi   amdsmi_get_gpu_metrics_info (current_socket_power, mean of current_gfxclks over the XCDs -- the per-XCD clocks are what DVFS moves)
           falling back to amdsmi_get_power_info / amdsmi_get_clock_info(GFX)
  sysfs    /sys/class/drm/card*/device/hwmon/hwmon*/power1_{input,average} (microwatt) and freq1_input (Hz)
  smi      `rocm-smi --showpower --showclocks --json` (slow: ~3 Hz at best; last resort)
The FIRST line of the file is a header {"header": ..} naming the source and the raw first reading, so a wrong field choice is visible."""
import argparse
import glob
import json
import os
import select
import signal
import subprocess
import sys
import time


def _num(v):
    try:
        f = float(v)
        return f if f == f and f not in (float('inf'), float('-inf')) else None
    except (TypeError, ValueError):
        return None


class AmdSmi:
    name = 'amdsmi'

    def __init__(self, device):
        import amdsmi
        self.m = amdsmi
        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        # HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES renumber devices for the bench process; amdsmi does not: honour the first entry
        vis = os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('ROCR_VISIBLE_DEVICES') or os.environ.get('CUDA_VISIBLE_DEVICES')
        idx = device
        if vis:
            try:
                idx = [int(x) for x in vis.split(',')][device]
            except (ValueError, IndexError):
                idx = device
        self.h = hs[idx if idx < len(hs) else 0]
        self.raw = None

    def read(self):
        m, out = self.m, {}
        try:
            g = m.amdsmi_get_gpu_metrics_info(self.h)
            if self.raw is None:
                self.raw = {k: (v if not isinstance(v, (list, tuple)) else list(v)[:10]) for k, v in g.items()
                            if any(s in k for s in ('power', 'gfxclk', 'socket', 'throttle', 'activity', 'temperature_hotspot'))}
            p = _num(g.get('current_socket_power'))
            if p is None or p <= 0 or p >= 65535:
                p = _num(g.get('average_socket_power'))
            if p is not None and 0 < p < 65535:
                out['power_w'] = p
            ck = [c for c in (_num(x) for x in (g.get('current_gfxclks') or [])) if c is not None and 0 < c < 65535]
            if ck:
                out['sclk_mhz'] = sum(ck) / len(ck)
            else:
                c = _num(g.get('current_gfxclk'))
                if c is not None and 0 < c < 65535:
                    out['sclk_mhz'] = c
        except Exception:                                   # noqa: BLE001  an SMI build without the metrics table: fall through to the narrower calls
            pass
        if 'power_w' not in out:
            try:
                pi = m.amdsmi_get_power_info(self.h)
                for k in ('current_socket_power', 'average_socket_power', 'socket_power'):
                    p = _num(pi.get(k))
                    if p is not None and p > 0:
                        out['power_w'] = p
                        break
            except Exception:                               # noqa: BLE001
                pass
        if 'sclk_mhz' not in out:
            try:
                ci = m.amdsmi_get_clock_info(self.h, m.AmdSmiClkType.GFX)
                c = _num(ci.get('clk', ci.get('cur_clk')))
                if c is not None and c > 0:
                    out['sclk_mhz'] = c
            except Exception:                               # noqa: BLE001
                pass
        return out


class Sysfs:
    name = 'sysfs'

    def __init__(self, device):
        cards = sorted(glob.glob('/sys/class/drm/card[0-9]*/device/hwmon/hwmon*'))
        cards = [c for c in cards if glob.glob(c + '/power1_*')]
        if not cards:
            raise RuntimeError('no amdgpu hwmon with a power sensor')
        self.d = cards[device if device < len(cards) else 0]
        self.pf = next(f for f in (self.d + '/power1_input', self.d + '/power1_average') if os.path.exists(f))
        self.ff = self.d + '/freq1_input' if os.path.exists(self.d + '/freq1_input') else None
        self.raw = {'dir': self.d, 'power_file': self.pf, 'freq_file': self.ff}

    def read(self):
        out = {}
        try:
            out['power_w'] = float(open(self.pf).read()) / 1e6
            if self.ff:
                out['sclk_mhz'] = float(open(self.ff).read()) / 1e6
        except (OSError, ValueError):
            pass
        return out


class SmiCli:
    name = 'smi'

    def __init__(self, device):
        self.dev = device
        self.raw = None
        if not self.read():
            raise RuntimeError('rocm-smi gave nothing')

    def read(self):
        out = {}
        try:
            r = subprocess.run(['rocm-smi', '-d', str(self.dev), '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=5)
            j = json.loads(r.stdout)
            card = next(iter(j.values()))
            if self.raw is None:
                self.raw = card
            for k, v in card.items():
                kl = k.lower()
                if 'sclk' in kl and 'clock' in kl and 'level' in kl:
                    s = str(v)
                    if '(' in s:                             # "3 (2100Mhz)": the level index comes first -- round 3 parsed that index as the clock
                        s = s[s.index('(') + 1:]
                    c = _num(s.lower().replace('mhz', '').replace(')', '').strip())
                    if c:
                        out['sclk_mhz'] = c
                if 'power' in kl and ('socket' in kl or 'average' in kl or 'current' in kl):
                    p = _num(v)
                    if p:
                        out['power_w'] = p
        except Exception:                                   # noqa: BLE001
            pass
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--device', type=int, default=0)
    ap.add_argument('--out', required=True)
    ap.add_argument('--hz', type=float, default=20.0)
    ap.add_argument('--seconds', type=float, default=0.0, help='stop by itself after this long (0 = until SIGTERM or stdin closes)')
    a = ap.parse_args()
    src, errs = None, {}
    for cls in (AmdSmi, Sysfs, SmiCli):
        try:
            s = cls(a.device)
            first = s.read()
            if 'power_w' in first or 'sclk_mhz' in first:
                src = s
                break
            errs[cls.name] = 'no power / clock field answered'
        except Exception as ex:                             # noqa: BLE001
            errs[cls.name] = repr(ex)[:200]
    stop = []
    signal.signal(signal.SIGTERM, lambda *_: stop.append(1))
    with open(a.out, 'w', buffering=1) as f:
        f.write(json.dumps({'header': {'source': src.name if src else None, 'errors': errs, 'raw_first': getattr(src, 'raw', None), 'hz': a.hz}}, default=str) + '\n')
        if src is None:
            return 1
        period, t_end = 1.0 / a.hz, (time.time() + a.seconds) if a.seconds else None
        nxt = time.time()
        while not stop and (t_end is None or time.time() < t_end):
            r = src.read()
            r['t'] = time.time()
            f.write(json.dumps(r) + '\n')
            nxt += period
            d = nxt - time.time()
            if d > 0:
                # sleep, but leave at once when the parent closes our stdin (it died or is done)
                try:
                    rl, _, _ = select.select([sys.stdin], [], [], d)
                    if rl and not sys.stdin.read(1):
                        break
                except (OSError, ValueError):
                    time.sleep(d)
            else:
                nxt = time.time()
    return 0


def summarise(path, t0, t1, peak_tflops=2500.0, max_mhz=2400.0):
    """Mean / min / max of the samples with t0 <= t <= t1 (time.time() stamps) -> the keys bench.py merges into `roofline`."""
    hdr, pw, ck = None, [], []
    try:
        with open(path) as f:
            for line in f:
                try:
                    r = json.loads(line)
                except ValueError:
                    continue
                if 'header' in r:
                    hdr = r['header']
                    continue
                if t0 <= r.get('t', 0) <= t1:
                    if 'power_w' in r:
                        pw.append(r['power_w'])
                    if 'sclk_mhz' in r:
                        ck.append(r['sclk_mhz'])
    except OSError as ex:
        return {'power_w_mean': None, 'sclk_mhz_mean': None, 'power_source': None, 'power_error': repr(ex)}
    out = {'power_source': (hdr or {}).get('source'), 'power_samples': len(pw), 'sclk_samples': len(ck),
           'power_w_mean': sum(pw) / len(pw) if pw else None, 'power_w_max': max(pw) if pw else None,
           'sclk_mhz_mean': sum(ck) / len(ck) if ck else None, 'sclk_mhz_min': min(ck) if ck else None, 'sclk_mhz_max': max(ck) if ck else None,
           'sclk_mhz_at_peak': max_mhz}
    if hdr and hdr.get('errors'):
        out['power_source_errors'] = hdr['errors']
    if ck:
        out['peak_at_sclk'] = peak_tflops * out['sclk_mhz_mean'] / max_mhz
    return out


if __name__ == '__main__':
    sys.exit(main())
