#!/usr/bin/env python3
"""Board power and clocks (hwmon) while one kernel mode runs back to back for a few seconds: is the chain kernel's
clock (s_memtime advances 1.5-1.67 G ticks/s inside it, 1.85-1.95 in lighter kernels) set by the power limit?
  python tools/power_probe.py [--modes idle,single,chain32] [--seconds 3]
VPTQ_HIP_LIB selects the build (ablated variants draw less)."""
import argparse, glob, os, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hwmon_dir():
    """the card whose PCI address is the one of HIP device 0 (a box of the pool shows every GPU of its node in /sys;
    only one of them is ours)"""
    import bench
    c = bench.sysfs_card_of_device(0)
    h = sorted(glob.glob(os.path.join(c, "hwmon", "hwmon*"))) if c else []
    return (c, h[0]) if h else (None, None)


def rd(path):
    try:
        return open(path).read().strip()
    except Exception:
        return None


class Sampler:
    def __init__(self, hw, names):
        self.paths = {n: os.path.join(hw, n) for n in names if os.path.exists(os.path.join(hw, n))}
        self.samples = {n: [] for n in self.paths}
        self._stop = False

    def __enter__(self):
        def loop():
            while not self._stop:
                for n, p in self.paths.items():
                    v = rd(p)
                    if v is not None:
                        try:
                            self.samples[n].append(float(v))
                        except ValueError:
                            pass
                time.sleep(0.01)
        self.t = threading.Thread(target=loop, daemon=True)
        self.t.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        self.t.join(timeout=1.0)

    def med(self, n, scale):
        s = sorted(self.samples.get(n, []))
        return None if not s else (s[len(s) // 2] * scale, s[0] * scale, s[-1] * scale)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="idle,single,chain32")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--hidden", type=int, default=8192)
    ap.add_argument("--exe", default="", help="instead of the modes: sample while this command runs (it should loop for a few seconds)")
    a = ap.parse_args()
    card, hw = hwmon_dir()
    p0 = torch.cuda.get_device_properties(0)
    print(f"HIP device 0 = PCI {p0.pci_domain_id:04x}:{p0.pci_bus_id:02x}:{p0.pci_device_id:02x} -> {card}")
    print("hwmon:", hw, sorted(os.listdir(hw)) if hw else None)
    if hw:
        for n in ("power1_cap", "power1_cap_max", "power1_cap_default", "power1_label"):
            v = rd(os.path.join(hw, n))
            if v is not None:
                print(f"  {n} = {v}")
    names = ["power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input", "in0_input"]
    if a.exe:
        import subprocess
        with Sampler(hw, names) as sm:
            time.sleep(0.3)
            r = subprocess.run(a.exe, shell=True, capture_output=True, text=True)
        print(r.stdout.strip())
        for nme, scale, unit in (("power1_input", 1e-6, "W"), ("freq1_input", 1e-6, "MHz")):
            s_ = sorted(sm.samples.get(nme, []))
            if s_:
                print(f"  {nme}: median {s_[len(s_) // 2] * scale:.0f}, p90 {s_[int(len(s_) * 0.9)] * scale:.0f}, max {s_[-1] * scale:.0f}, min {s_[0] * scale:.0f} {unit} ({len(s_)} samples)")
        return
    import bench
    from vptq_amd.ops.chain import GemvChain
    from vptq_amd import _backend as B
    dev = torch.device("cuda", 0)
    I = a.hidden
    g = torch.Generator(device=dev).manual_seed(1)
    ring = [bench.make_layer(I, I, dev, g) for _ in range(32)]
    x = torch.randn(1, 1, I, device=dev, generator=g).half()
    ys = [torch.empty(1, 1, I, dtype=torch.float16, device=dev) for _ in range(32)]

    def run_single():
        for m, y in zip(ring, ys):
            d = m._descriptor()
            B.check(d[4](d[1], x.data_ptr(), y.data_ptr(), 1, 0, None, 0, B.current_stream_ptr(dev)), "gemv")

    chain = GemvChain(ring)

    def run_chain():
        chain([x] * 32, ys)

    big = torch.empty(1 << 28, dtype=torch.float16, device=dev)

    def run_stream():   # a plain device-to-device reduction: 512 MiB read per call
        big.sum()

    fns = {"single": run_single, "chain32": run_chain, "stream": run_stream}
    for mode in a.modes.split(","):
        graph = None
        if mode != "idle":
            fn = fns[mode]
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                fn()
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=s):
                    for _ in range(8):
                        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        with Sampler(hw, names) as sm:
            t0 = time.time()
            e0.record()
            while time.time() - t0 < a.seconds:
                if graph is not None:
                    for _ in range(20):
                        graph.replay()
                    n += 20 * 8
                    torch.cuda.synchronize()
                else:
                    time.sleep(0.05)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        line = f"{mode:8s}"
        if n:
            line += f" {ms * 1e3 / n / (32 if mode != 'stream' else 1):8.2f} us per " + ("layer" if mode != "stream" else "512 MiB pass")
        for nme, scale, unit in (("power1_average", 1e-6, "W"), ("power1_input", 1e-6, "W"), ("freq1_input", 1e-6, "MHz"),
                                 ("freq2_input", 1e-6, "MHz"), ("temp1_input", 1e-3, "C"), ("temp2_input", 1e-3, "C"),
                                 ("in0_input", 1.0, "mV")):
            r = sm.med(nme, scale)
            if r:
                line += f" | {nme} median {r[0]:.0f} (min {r[1]:.0f}, max {r[2]:.0f}) {unit}, {len(sm.samples[nme])} samples"
        print(line, flush=True)


if __name__ == "__main__":
    main()
