"""`--profile <dir>` of the entry scripts (the reference has only tqdm, gan_training.py:380): a per-kernel table of N
training steps of each phase, from the same HIP-event wrappers bench.py's roofline pass uses (hip.start_profile), plus roctx
ranges around every step so that `rocprofv3 --marker-trace --kernel-trace -- python train_gan.py ...` groups its kernels by step.

    prof = StepProfiler(dir, steps=3, skip=2, rank=0)      # or StepProfiler.off()
    with prof.step("phase2"): <one training step>          # in the training loops (training.py, auto_training.py)

For every phase name the first `skip` steps pass untimed, the next `steps` are profiled, then <dir>/kernels_<phase>.txt (a table
sorted by time) and .json are written and the wrappers removed: the run continues at full speed."""
import ctypes
import json
import os

_roctx = None


def _roctx_lib():
    global _roctx
    if _roctx is None:
        _roctx = False
        for name in ("libroctx64.so", "librocprofiler-sdk-roctx.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                _roctx = lib
                break
            except (OSError, AttributeError):
                continue
    return _roctx


class _Step(object):
    def __init__(self, prof, phase):
        self.prof, self.phase = prof, phase

    def __enter__(self):
        self.prof._enter(self.phase)
        return self

    def __exit__(self, *exc):
        self.prof._exit(self.phase, exc[0] is not None)
        return False


class _Off(object):
    """No --profile: step() costs one attribute lookup."""
    enabled = False
    active = None

    def step(self, phase):
        return self

    def begin(self, phase):
        pass

    def end(self, phase, failed=False):
        pass

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class StepProfiler(object):
    enabled = True

    def __init__(self, out_dir, steps=3, skip=2, rank=0):
        self.dir, self.steps, self.skip, self.rank = out_dir, max(1, int(steps)), max(0, int(skip)), rank
        self.seen, self.done, self.active = {}, set(), None
        if rank == 0:
            os.makedirs(out_dir, exist_ok=True)

    @staticmethod
    def off():
        return _Off()

    def step(self, phase):
        return _Step(self, phase)

    def begin(self, phase):
        self._enter(phase)

    def end(self, phase, failed=False):
        self._exit(phase, failed)

    def _enter(self, phase):
        r = _roctx_lib()
        if r:
            r.roctxRangePushA(("%s step %d" % (phase, self.seen.get(phase, 0))).encode())
        if phase in self.done or self.rank != 0:
            return
        if self.active is not None and self.active != phase:
            # the previous phase ended before its quota of steps (train_gan.py --phase1_steps 3): write what it got
            self._finish(self.active, False)
        n = self.seen.get(phase, 0)
        if n == self.skip and self.active is None:
            from . import hip
            hip.start_profile()
            self.active = phase

    def _exit(self, phase, failed):
        r = _roctx_lib()
        if r:
            r.roctxRangePop()
        n = self.seen[phase] = self.seen.get(phase, 0) + 1
        if self.active == phase and (failed or n >= self.skip + self.steps):
            self._finish(phase, failed)

    def _finish(self, phase, failed):
        from . import hip
        agg = hip.stop_profile().summary()
        self.active = None
        self.done.add(phase)
        steps = self.seen.get(phase, 0) - self.skip
        if not failed and steps > 0:
            self._write(phase, agg, steps)

    def close(self):
        """End of the run: a phase still being profiled is written with the steps it got."""
        if self.active is not None:
            self._finish(self.active, False)

    def _write(self, phase, agg, steps):
        rows = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
        total = sum(v["ms"] for _, v in rows) or 1.0
        with open(os.path.join(self.dir, "kernels_%s.txt" % phase), "w") as f:
            f.write("# %s: %d profiled step(s); HIP-event time per launcher call (recguru_amd.hip.start_profile), single stream order\n" % (phase, steps))
            f.write("%-46s %9s %11s %10s %7s %10s %10s\n" % ("kernel", "launches", "ms/step", "avg us", "share", "TFLOP/s", "GB/s"))
            for k, v in rows:
                sec = max(v["ms"], 1e-9) * 1e-3
                f.write("%-46s %9.1f %11.3f %10.1f %6.1f%% %10.1f %10.1f\n" % (k[:46], v["launches"] / steps, v["ms"] / steps,
                        v["ms"] * 1e3 / max(v["launches"], 1), 100.0 * v["ms"] / total, v["flops_exec"] / sec / 1e12, v["bytes_exec"] / sec / 1e9))
            f.write("%-46s %9.1f %11.3f\n" % ("total", sum(v["launches"] for _, v in rows) / steps, total / steps))
        with open(os.path.join(self.dir, "kernels_%s.json" % phase), "w") as f:
            # every figure per profiled step: launches, ms, algorithmic and executed flops / bytes
            json.dump({"phase": phase, "steps": steps, "kernels": {k: {kk: vv / steps for kk, vv in v.items()} for k, v in rows}}, f, indent=1)


_CURRENT = _Off()


def install(prof):
    """The profiler the training loops consult (training.train_recon_x / train_gan_all / recommendation_tune, auto_training.train)."""
    global _CURRENT
    _CURRENT = prof if prof is not None else _Off()
    return _CURRENT


def current():
    return _CURRENT
