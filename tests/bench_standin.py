"""TEST INFRASTRUCTURE -- a CPU stand-in for the GPU side of bench.py, so that bench.main()'s N > 1 path (self-launch of the
ranks, env sharding, the per-range launch / restart / gather schedule, fences, max-over-ranks reduction, the parity rows of
every rank reaching rank 0 under their global env ids, ONE JSON line from rank 0) can be driven end to end on a box without
GPUs (`bench.py --dry-run-cpu`, tests/test_multirank.py).  No physics is computed and nothing is measured: a "step" adds
nsub * (the bound PD targets rounded to 1/1024: every sum is exact in fp64, whatever the launch boundaries) to the first ten
qpos columns, and the "CPU reference" replays that arithmetic -- so the stand-in's qpos error is exactly zero unless a shard
is mislabelled, a restart hits the wrong rows or a gather is out of global env order.
Never imported by the product or by a GPU run of bench.py."""
import contextlib
import ctypes
import time

import numpy as np
import torch


def _view(ptr, rows, cols, row_stride=None):
    """numpy view of host memory a CPU tensor's data_ptr() points at (the stand-in works on raw pointers like the library)."""
    st = cols if row_stride is None else row_stride
    flat = np.ctypeslib.as_array((ctypes.c_double * (rows * st)).from_address(ptr))
    return flat.reshape(rows, st)[:, :cols]


class Stream:
    cuda_stream = 0

    def wait_event(self, ev):
        pass


class Event:
    def __init__(self, enable_timing=False):
        self.t = None

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return 1e3 * (other.t - self.t)


class StandInBatch:
    """The subset of cassie_amd.Batch that bench.device_rollout uses."""

    def __init__(self, model, n):
        from cassie_amd import phys as P
        self.P, self.pod, self.nenv = P, model.pod, n
        self.ptr, self.stride = {}, {}
        self.launches = 0

    # configuration calls of the real batch that mean nothing here
    def set_all_outputs_every_substep(self, on=True): pass
    def set_fast_rows(self, on=True): pass
    def set_waves_per_env(self, waves=2): pass
    def set_balance(self, on=True): pass
    def set_hfield(self, h): pass
    def set_drive_mode(self, mode): pass
    def set_pd_mode(self, on=True): pass
    def enable_kernel_timing(self, on=True): pass
    def close(self): pass

    def bind(self, field, ptr, row_stride=None):
        self.ptr[field], self.stride[field] = ptr, row_stride

    def _qpos(self):
        return _view(self.ptr[self.P.F_QPOS], self.nenv, self.pod.nq, self.stride[self.P.F_QPOS])

    def step_range(self, first, cnt, nsub, stream=None):
        tg = _view(self.ptr[self.P.F_PD_PTARGET], self.nenv, 10)
        self._qpos()[first:first + cnt, :10] += nsub * (np.round(tg[first:first + cnt] * 1024.0) / 1024.0)
        self.launches += 1

    def step(self, nsub, stream=None):
        self.step_range(0, self.nenv, nsub, stream)

    def reset_envs(self, r0, stride, k, init_ptr, sens_ptr, stream=None):
        nobs = self.stride[self.P.F_QPOS]
        init = _view(init_ptr, 1, nobs)[0]
        rows = _view(self.ptr[self.P.F_QPOS], self.nenv, nobs, nobs)
        rows[r0:r0 + stride * k:stride] = init
        rows[r0:r0 + stride * k:stride, :10] = 0.0        # (a dyadic base: the sums of the stand-in's "steps" stay exact)

    def kernel_timing(self):
        n, self.launches = self.launches, 0
        return n, 0.0

    def warnings(self):
        return np.zeros(self.nenv, dtype=np.int32), np.zeros((self.nenv, 4), dtype=np.int32)

    def fast_rows_progress(self):
        return np.full(self.nenv, 1 << 20, dtype=np.int32)


class StandInEnvs:
    """The "CPU reference" of the stand-in: the same arithmetic for the sampled envs, through bench.Schedule."""

    def __init__(self, model, env_ids, hfield=None):
        import bench
        self.bench, self.ids, self.q0 = bench, np.asarray(env_ids), model.qpos_init()
        self.q = np.tile(self.q0, (len(self.ids), 1))

    def restart(self, group):
        self.q[self.ids % self.bench.NGROUP == group] = self.q0
        self.q[self.ids % self.bench.NGROUP == group, :10] = 0.0

    def step(self, nsub, targets, threads=1):
        self.q[:, :10] += nsub * (np.round(np.asarray(targets) * 1024.0) / 1024.0)

    def qpos(self):
        return self.q.copy()

    def counts(self):
        return np.zeros((len(self.ids), 3), dtype=np.int64)


class CpuRuntime:
    """What bench.device_rollout asks of its runtime (see bench.GpuRuntime), on the CPU with gloo."""
    backend = "gloo"
    name = "stand-in (CPU, no physics: launch-path test)"

    def __init__(self, local_rank):
        self.device = torch.device("cpu")

    Stream = staticmethod(lambda: Stream())
    Event = staticmethod(lambda enable_timing=False: Event(enable_timing))

    def use(self, stream):
        return contextlib.nullcontext()

    def synchronize(self):
        pass

    def make_batch(self, model, n):
        return StandInBatch(model, n)

    def init_sensordata(self, model, hfield):
        return np.zeros(model.pod.nsensordata)

    def replay_envs(self, drive):
        return StandInEnvs

    def host_threads(self):
        return 1
