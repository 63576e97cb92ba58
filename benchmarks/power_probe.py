"""Socket power and shader clock while one kernel runs back to back: is the wide conv at the chip's power budget?

    python benchmarks/power_probe.py [--seconds 4] [--out gpurun_out/power_probe.json]

For each (shape, operand fill) a program of identical launches runs for a few seconds on the stream while a host thread samples
what the box exposes: the amdgpu hwmon files (power1_average / power1_input in uW, power1_cap, freq1_input in Hz), and, at a
lower rate, `amd-smi metric` / `rocm-smi`.  The same binary on zero operands is the control: same cycles, less switching.
"""
import argparse
import glob
import json
import math
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from img2img_turbo_amd import _capi as K, ops as O  # noqa: E402

SHAPES = [("vae 128->128@512 gn", 128, 128, 512), ("vae 512->512@128 gn", 512, 512, 128), ("unet lin 1280->10240 T256 x64", 1280, 10240, 0)]


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def hwmon_files():
    out = {}
    for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for n in ("power1_average", "power1_input", "power1_cap", "power1_cap_max", "freq1_input", "freq2_input", "temp1_input", "in0_input"):
            p = os.path.join(h, n)
            if os.path.exists(p):
                out.setdefault(n, p)
    return out


def smi_numbers(raw):
    """(socket power W, mean gfx clock MHz over the XCDs) out of one `amd-smi metric --power --clock --json` reading (text search: the
    stored reading may be truncated)."""
    import re
    m = re.search(r'"socket_power":\s*\{\s*"value":\s*(\d+)', raw)
    c = [int(v) for v in re.findall(r'"gfx_\d+":\s*\{\s*"clk":\s*\{\s*"value":\s*(\d+)', raw)]
    return (int(m.group(1)) if m else None), (round(sum(c) / len(c)) if c else None)


def smi_once():
    """One slow sample through the command-line tools (whatever this image has); returns a short dict of strings."""
    for cmd in (["amd-smi", "metric", "-g", "0", "--power", "--clock", "--json"], ["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--json"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
        except (OSError, subprocess.TimeoutExpired):
            continue
        if r.returncode == 0 and r.stdout.strip():
            return {"cmd": " ".join(cmd[:2]), "raw": r.stdout.strip()[:4000]}
    return None


class Sampler(threading.Thread):
    def __init__(self, files, smi):
        super().__init__(daemon=True)
        self.files, self.smi, self.rows, self.smi_rows, self.stop = files, smi, [], [], False

    def run(self):
        t_smi = 0.0
        while not self.stop:
            row = {"t": time.time()}
            for n, p in self.files.items():
                v = _read(p)
                if v is not None and v.lstrip("-").isdigit():
                    row[n] = int(v)
            self.rows.append(row)
            if self.smi and time.time() - t_smi > 1.0:
                s = smi_once()
                t_smi = time.time()
                if s:
                    self.smi_rows.append(s)
            time.sleep(0.02)


def stats(rows, key, scale):
    v = [r[key] * scale for r in rows if key in r]
    if not v:
        return None
    v = v[len(v) // 5:]                      # drop the ramp at the start
    return {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "n": len(v)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--out", default="gpurun_out/power_probe.json")
    ap.add_argument("--no-smi", action="store_true")
    a = ap.parse_args()
    dt = torch.bfloat16
    lib = K.default_library()
    files = hwmon_files()
    print("hwmon files:", {k: v for k, v in files.items()}, flush=True)
    for n in ("power1_cap", "power1_cap_max"):
        if n in files:
            print("%s = %s uW" % (n, _read(files[n])), flush=True)
    first = None if a.no_smi else smi_once()
    print("smi tool:", (first or {}).get("cmd"), flush=True)
    if first:
        print("idle smi sample:", first["raw"][:1500].replace("\n", " "), flush=True)
    idle = Sampler(files, False)
    idle.start(); time.sleep(1.0); idle.stop = True; idle.join()
    res = {"idle": {k: stats(idle.rows, k, s) for k, s in (("power1_average", 1e-6), ("power1_input", 1e-6), ("freq1_input", 1e-6))}}
    print("idle:", res["idle"], flush=True)
    B = a.batch
    for name, cin, cout, hw in SHAPES:
        for fill in ("randn", "zero", "randn"):
            if hw:
                x = torch.randn(B, hw, hw, cin, device="cuda").to(dt)
                w = (torch.randn(cout, 9 * cin, device="cuda") / math.sqrt(9 * cin)).to(dt)
                ss = torch.randn(B, cin, 2, device="cuda")
                ks, nimg, H = 3, B, hw
                fl = 2.0 * B * hw * hw * cout * 9 * cin
            else:
                x = torch.randn(B * 64, 16, 16, cin, device="cuda").to(dt)      # 64 x the T=256 rows of one batch-8 launch: long enough to fill the chip
                w = (torch.randn(cout, cin, device="cuda") / math.sqrt(cin)).to(dt)
                ss, ks, nimg, H = None, 1, B * 64, 16
                fl = 2.0 * nimg * 256 * cout * cin
            if fill == "zero":
                x.zero_(); w.zero_()
            out = torch.empty(nimg, H, H, cout, device="cuda", dtype=dt)
            bias = torch.randn(cout, device="cuda")

            def program(n):
                prog = K.Program()
                for _ in range(n):
                    op = O.conv(x, w, out, nimg=nimg, hin=H, win=H, ho=H, wo=H, ks=ks, stride=1, pad=ks // 2, ups=0, N=cout, gn_ss=ss,
                                act=1 if ss is not None else 0, bias=bias, tile=0, splitk=0, ws=None, subpix=0, res=None, ldr=None, geglu=0, ldc=cout)
                    prog.add(op[0], O.DT[dt], op[1])
                prog.freeze()
                return prog
            st = torch.cuda.current_stream().cuda_stream
            ms0 = sorted(lib.run_timed(program(6), st)[1:])[2]
            n = max(8, min(20000, int(a.seconds * 1e3 / ms0)))
            prog = program(n)
            smp = Sampler(files, not a.no_smi)
            smp.start()
            ms = lib.run_timed(prog, st)
            smp.stop = True; smp.join()
            k = len(ms) // 5
            tail = sorted(ms[k:])
            t_med, t_first = tail[len(tail) // 2], sorted(ms[:max(3, k // 4)])[max(1, k // 8)]
            rec = {"shape": name, "fill": fill, "launches": n, "ms_median_after_ramp": t_med, "ms_first_launches": t_first, "tflops": fl / t_med / 1e9,
                   "power_w": stats(smp.rows, "power1_average", 1e-6) or stats(smp.rows, "power1_input", 1e-6),
                   "sclk_mhz": stats(smp.rows, "freq1_input", 1e-6), "smi": smp.smi_rows[-2:]}      # (hwmon of card0 may be another device: amd-smi is the reading)
            res.setdefault("runs", []).append(rec)
            pw, ck = rec["power_w"], rec["sclk_mhz"]
            print("%-30s %-5s  %6d launches  %.4f ms (first launches %.4f)  %7.1f TF   power %s W   sclk %s MHz" % (
                name, fill, n, t_med, t_first, rec["tflops"],
                ("%.0f (%.0f..%.0f)" % (pw["mean"], pw["min"], pw["max"])) if pw else "n/a",
                ("%.0f (%.0f..%.0f)" % (ck["mean"], ck["min"], ck["max"])) if ck else "n/a"), flush=True)
            if smp.smi_rows:
                nums = [smi_numbers(r["raw"]) for r in smp.smi_rows]
                rec["smi_socket_w"], rec["smi_gfx_mhz"] = [n[0] for n in nums], [n[1] for n in nums]
                print("   amd-smi: socket power %s W   gfx clock %s MHz (mean over the XCDs; one reading per second)" % (rec["smi_socket_w"], rec["smi_gfx_mhz"]), flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
