#!/usr/bin/env python3
"""Round 6, VERDICT r5 weak #4 / task 3(i): why does the SAME wide SDF kernel run at 1.8-1.9 ns per point in a frame's 16-sample
passes (2.1 M points per launch) and at 2.5 ns per point in its 64-sample passes (8.4 M points)?

Per kernel mode (0 = sdf, 1 = + gradient, 2 = + feature) and launch size this prints ns per point
  cold       ONE launch after 300 ms of idle GPU            (boost clock, cold power state)
  sustained  the mean of 8 back-to-back launches             (what a frame's long passes see)
  paced      launches separated by idle gaps as long as themselves (duty cycle 50 %)
with the shader clock and package power polled from sysfs (hwmon freq1_input / power1_average) every ~1 ms beside them, and
  geometry   the same number of points as 16 samples x many rays and as 64 samples x a quarter of the rays, and at points
             clustered near the surface (importance samples) against points spread along the ray (coarse samples)
so that the three candidate causes - the power governor, a tail / tile-scheduling effect, the data - can be told apart.

    python profiles/launch_length_probe.py [out.json]
"""
import glob
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na  # noqa: E402
from nrhints_amd import ops  # noqa: E402
from nrhints_amd.synthetic import make_rays, perturb_state  # noqa: E402


class SmiPoller:
    """sclk [MHz] and package power [W] from the amdgpu hwmon files, polled in a thread (rocm-smi itself takes ~100 ms per call)."""

    def __init__(self):
        self.freq = self.power = None
        want = None
        try:       # the card torch runs on (a box shows all eight cards of the node in sysfs; only one is visible to HIP)
            pr = torch.cuda.get_device_properties(0)
            want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:  # noqa: BLE001
            pass
        cards = []
        for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
            slot = ""
            try:
                for line in open(os.path.join(dev, "uevent")):
                    if line.startswith("PCI_SLOT_NAME="):
                        slot = line.strip().split("=", 1)[1]
            except Exception:  # noqa: BLE001
                pass
            for h in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
                cards.append((slot, h))
        self.card = None
        for slot, h in cards:
            if want is not None and slot.lower() != want.lower():
                continue
            if os.path.exists(os.path.join(h, "freq1_input")):
                self.freq = os.path.join(h, "freq1_input")
            for name in ("power1_average", "power1_input"):
                if os.path.exists(os.path.join(h, name)) and self.power is None:
                    self.power = os.path.join(h, name)
            if self.freq or self.power:
                self.card = slot
                break
        print("visible device", want, "-> hwmon of", self.card, "(of", len(cards), "cards in sysfs)", flush=True)
        self.samples, self._stop, self._t = [], False, None

    def _read(self, path, scale):
        try:
            with open(path) as f:
                return float(f.read().strip()) * scale
        except Exception:  # noqa: BLE001
            return float("nan")

    def start(self):
        self.samples, self._stop = [], False

        def loop():
            while not self._stop:
                self.samples.append((time.perf_counter(), self._read(self.freq, 1e-6) if self.freq else float("nan"),
                                     self._read(self.power, 1e-6) if self.power else float("nan")))
                time.sleep(0.001)
        self._t = threading.Thread(target=loop, daemon=True)
        self._t.start()

    def stop(self):
        self._stop = True
        self._t.join()
        return self.samples

    def window(self, t0, t1):
        s = [x for x in self.samples if t0 <= x[0] <= t1]
        if not s:
            return None
        f, p = np.array([x[1] for x in s]), np.array([x[2] for x in s])
        return dict(n=len(s), sclk_mhz_mean=float(np.nanmean(f)), sclk_mhz_min=float(np.nanmin(f)), sclk_mhz_max=float(np.nanmax(f)),
                    power_w_mean=float(np.nanmean(p)), power_w_max=float(np.nanmax(p)))


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    model = na.NeuSHintRenderer(na.NeuSModelConfig(), precision="f16x3")
    st = perturb_state({k: v.detach().numpy().copy() for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    pk = model.to(dev).eval().packed_params(dev)
    w32, tab = pk["sdf_w32"], pk["sdf_tab32"]
    nr_max = 131072
    o, d, pl, near, far = (torch.from_numpy(a).to(dev) for a in make_rays(nr_max, seed=1, spread=0.1))
    lin = torch.linspace(0, 1, 128, device=dev)[None]
    t_coarse = (near + (far - near) * lin).contiguous()                      # spread along the ray (coarse samples)
    mid = -(o * d).sum(-1, keepdim=True) - 0.5                                 # ~ the surface of the radius-0.5 sphere
    t_near = (mid + 0.05 * (lin - 0.5)).contiguous()                          # clustered near the surface (importance samples)
    scratch = ops._scratch(dev)
    poll = SmiPoller()
    print("hwmon:", poll.freq, poll.power, flush=True)
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def launch(mode, nrays, per_ray, tt):
        return ops.sdf_eval_wide(mode, w32, tab, o[:nrays], d[:nrays], tt[:nrays], per_ray, t_stride=128, scratch=scratch)

    def timed(mode, nrays, per_ray, tt, reps, gap_s):
        """-> ([ms per launch], (t0, t1) host window of the launches)"""
        launch(mode, nrays, per_ray, tt)
        torch.cuda.synchronize()
        time.sleep(0.3)
        pairs = []
        t0 = time.perf_counter()
        for _ in range(reps):
            a, b = ev(), ev()
            a.record()
            launch(mode, nrays, per_ray, tt)
            b.record()
            pairs.append((a, b))
            if gap_s:
                torch.cuda.synchronize()
                time.sleep(gap_s)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        return [a.elapsed_time(b) for a, b in pairs], (t0, t1)

    rows = []
    poll.start()
    try:
        for mode in (0, 1, 2):
            for nrays, per_ray, geom in ((16384, 16, "coarse"), (131072, 16, "coarse"), (131072, 16, "near"), (32768, 64, "coarse"),
                                         (32768, 64, "near"), (131072, 64, "coarse"), (131072, 64, "near"), (65536, 128, "near"),
                                         (131072, 128, "near"), (131072, 128, "coarse")):
                tt = t_coarse if geom == "coarse" else t_near
                npts = nrays * per_ray
                cold, wc = timed(mode, nrays, per_ray, tt, 1, 0.0)
                sus, ws = timed(mode, nrays, per_ray, tt, 8, 0.0)
                paced, wp = timed(mode, nrays, per_ray, tt, 6, max(1e-3, np.mean(sus) * 1e-3))
                row = dict(mode=mode, nrays=nrays, per_ray=per_ray, geometry=geom, mpts=round(npts / 1e6, 3),
                           cold_ns_pt=round(cold[0] * 1e6 / npts, 3), sustained_ns_pt=round(float(np.mean(sus)) * 1e6 / npts, 3),
                           sustained_first_last=[round(sus[0] * 1e6 / npts, 3), round(sus[-1] * 1e6 / npts, 3)],
                           paced_ns_pt=round(float(np.mean(paced)) * 1e6 / npts, 3), ms_sustained=round(float(np.mean(sus)), 3),
                           smi_cold=poll.window(*wc), smi_sustained=poll.window(*ws), smi_paced=poll.window(*wp))
                rows.append(row)
                s = row["smi_sustained"] or {}
                c = row["smi_cold"] or {}
                print(f"mode {mode} {nrays:6d} rays x {per_ray:3d} ({geom:6s}) {row['mpts']:7.3f} Mpts: cold {row['cold_ns_pt']:.3f} sustained "
                      f"{row['sustained_ns_pt']:.3f} (first {row['sustained_first_last'][0]:.3f} last {row['sustained_first_last'][1]:.3f}) paced "
                      f"{row['paced_ns_pt']:.3f} ns/pt | sclk cold {c.get('sclk_mhz_mean', float('nan')):.0f} sustained {s.get('sclk_mhz_mean', float('nan')):.0f} MHz, "
                      f"power sustained {s.get('power_w_mean', float('nan')):.0f} W (max {s.get('power_w_max', float('nan')):.0f})", flush=True)
    finally:
        poll.stop()
    if out_path:
        with open(out_path, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
