"""Engine: one libmcle context (one GPU, one stream) with device arrays and typed wrappers.

Everything here is plumbing around the C ABI (include/mcle.h); the arithmetic runs in the HIP
kernels under pyphysim_amd/csrc.  NumPy arrays passed to an operator are copied to the device
and the result is copied back (drop-in behaviour, PCIe-bound); :class:`DeviceArray` arguments
stay resident and results come back as DeviceArray.
"""
import contextlib
import ctypes
import math
from ctypes import byref, c_double, c_float, c_int, c_int32, c_void_p

import numpy as np

from . import _lib
from ._lib import (CONST_BPSK, CONST_GENERIC, CONST_QAM, DEMOD_MINDIST, DEMOD_QAM_SLICER, MCLE_F32, MCLE_F64,
                   AwgnCfg, Counters, FlatCfg, IaCfg, LegacySeg, McleError, MimoOfdmCfg, OfdmTdlCfg, check)


class DeviceArray:
    """A typed, shaped view of device memory owned by an Engine."""

    def __init__(self, engine, shape, dtype):
        self.engine = engine
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        self.nbytes = self.size * self.dtype.itemsize
        self._ptr = c_void_p(0)
        if self.nbytes:
            self._ptr = engine._alloc(self.nbytes)

    @property
    def ptr(self):
        return self._ptr

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        view = DeviceArray.__new__(DeviceArray)
        view.engine, view.dtype, view.size, view.nbytes = self.engine, self.dtype, self.size, self.nbytes
        shape = [int(v) for v in shape]
        if shape.count(-1) > 1:
            raise ValueError("can only specify one unknown dimension")
        if -1 in shape:
            known = int(np.prod([v for v in shape if v != -1])) if len(shape) > 1 else 1
            if known == 0 or self.size % known:
                raise ValueError("cannot reshape array of size %d into shape %s" % (self.size, tuple(shape)))
            shape[shape.index(-1)] = self.size // known
        if int(np.prod(shape)) != self.size:
            raise ValueError("cannot reshape array of size %d into shape %s" % (self.size, tuple(shape)))
        view.shape = tuple(shape)
        view._ptr, view._base = self._ptr, self
        return view

    def get(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            check(self.engine.lib.mcle_memcpy_d2h(self.engine.ctx, out.ctypes.data_as(c_void_p), self._ptr,
                                                  self.nbytes))
        return out

    def set(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.size != self.size:
            raise ValueError("size mismatch: %d vs %d" % (host.size, self.size))
        if self.nbytes:
            check(self.engine.lib.mcle_memcpy_h2d(self.engine.ctx, self._ptr, host.ctypes.data_as(c_void_p),
                                                  self.nbytes))
        return self

    def zero(self):
        if self.nbytes:
            check(self.engine.lib.mcle_memset(self.engine.ctx, self._ptr, 0, self.nbytes))
        return self

    def __del__(self):
        try:
            if getattr(self, "_base", None) is None and self._ptr and self._ptr.value and self.engine.ctx:
                self.engine._release(self._ptr, self.nbytes)
                self._ptr = c_void_p(0)
        except Exception:
            pass


class Engine:
    """One context on one MI355X.  ``dtype`` ('f64' parity / 'f32' throughput) is the default
    arithmetic of the operator calls; every call can override it."""

    def __init__(self, device=0, dtype="f64"):
        self.lib = _lib.load()
        self.ctx = c_void_p(0)
        self.dtype = _lib.dtype_code(dtype)
        ctx = c_void_p(0)
        check(self.lib.mcle_ctx_create(int(device), byref(ctx)))
        self.ctx = ctx
        self.device = int(device)
        self.M = 0
        self._table_key = None
        # size-keyed free list: operator calls allocate their outputs, and hipMalloc / hipFree of large
        # buffers cost far more than the kernels; all work is on one stream, so reuse is stream-ordered
        self._pool, self._pool_bytes, self.pool_limit = {}, 0, 8 << 30
        n_cu, lds = c_int(0), c_int(0)
        name = ctypes.create_string_buffer(128)
        check(self.lib.mcle_ctx_device_info(self.ctx, byref(n_cu), byref(lds), name, 128))
        self.n_cu, self.device_name = n_cu.value, name.value.decode()

    def _alloc(self, nbytes):
        free = self._pool.get(nbytes)
        if free:
            self._pool_bytes -= nbytes
            return free.pop()
        ptr = c_void_p(0)
        rc = self.lib.mcle_malloc(self.ctx, nbytes, byref(ptr))
        if rc and self._pool_bytes:
            self.empty_pool()
            rc = self.lib.mcle_malloc(self.ctx, nbytes, byref(ptr))
        check(rc)
        return ptr

    def _release(self, ptr, nbytes):
        if self._pool_bytes + nbytes <= self.pool_limit:
            self._pool.setdefault(nbytes, []).append(ptr)
            self._pool_bytes += nbytes
        else:
            self.lib.mcle_free(self.ctx, ptr)

    def empty_pool(self):
        """Returns the pooled device buffers and the context's scratch buffer (the record buffers of the two-launch
        pipelines, up to 138-320 MiB) to the allocator."""
        for free in self._pool.values():
            for ptr in free:
                self.lib.mcle_free(self.ctx, ptr)
        self._pool, self._pool_bytes = {}, 0
        check(self.lib.mcle_ctx_trim_scratch(self.ctx))

    def close(self):
        if self.ctx:
            self.empty_pool()
            self.lib.mcle_ctx_destroy(self.ctx)
            self.ctx = c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- plumbing ---------------------------------------------------------------------------
    def sync(self):
        check(self.lib.mcle_ctx_sync(self.ctx))

    # ---- kernel-selection options (mcle_ctx_set_option: per context, never read from the environment) ----
    def set_option(self, name, value):
        """name: a key of _lib.OPTIONS ('no_mfma', 'mfma_variant', 'grid_oversub', 'flat_wgs_per_cu', 'single_tdl',
        'tdl_mfma_waves', 'jakes_direct', 'f64_generic', 'f64_threads', 'bd_runtime_solve', 'demod_nocert', 'f64_variant', 'f32_mfma', 'tdl_kernel', 'mimo_tdl_kernel', 'walk_legacy':
        include/mcle.h MCLE_OPT_* says what each selects); 0 restores the default."""
        if name not in _lib.OPTIONS:
            raise ValueError("unknown option %r (known: %s)" % (name, ", ".join(sorted(_lib.OPTIONS))))
        check(self.lib.mcle_ctx_set_option(self.ctx, _lib.OPTIONS[name], int(value)))

    def get_option(self, name):
        if name not in _lib.OPTIONS:
            raise ValueError("unknown option %r" % (name,))
        v = ctypes.c_longlong(0)
        check(self.lib.mcle_ctx_get_option(self.ctx, _lib.OPTIONS[name], byref(v)))
        return int(v.value)

    @contextlib.contextmanager
    def options(self, **kw):
        """with eng.options(no_mfma=1): ...  -- sets the options for the block and restores the previous values."""
        old = {k: self.get_option(k) for k in kw}
        try:
            for k, v in kw.items():
                self.set_option(k, v)
            yield self
        finally:
            for k, v in old.items():
                self.set_option(k, v)

    def use_stream(self, hip_stream_handle):
        """Adopt an external hipStream_t, e.g. ``torch.cuda.current_stream().cuda_stream``."""
        check(self.lib.mcle_ctx_set_stream(self.ctx, c_void_p(int(hip_stream_handle) if hip_stream_handle else 0)))

    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype):
        return DeviceArray(self, shape, dtype).zero()

    def to_device(self, host, dtype=None):
        host = np.ascontiguousarray(host, dtype=dtype)
        return DeviceArray(self, host.shape, host.dtype).set(host)

    HBM_KINDS = {"copy": 0, "read": 1, "triad": 2, "write": 3}

    def hbm_stream_rate(self, kind="copy", nbytes=1 << 30, reps=10, blocks_per_cu=8, nontemporal=False, unroll8=False):
        """GB/s (bytes read + written per second) of the library's 16-byte streaming kernel over arrays of `nbytes` bytes:
        what this box's HBM delivers (mcle_hbm_stream_rate, csrc/kernels_hbm.hip)."""
        g = ctypes.c_double(0.0)
        check(self.lib.mcle_hbm_stream_rate(self.ctx, self.HBM_KINDS[kind] | (4 if nontemporal else 0) | (8 if unroll8 else 0), int(nbytes), int(reps), int(blocks_per_cu), byref(g)))
        return float(g.value)

    def timer_start(self):
        check(self.lib.mcle_timer_start(self.ctx))

    def timer_stop_ms(self):
        ms = c_float(0)
        check(self.lib.mcle_timer_stop_ms(self.ctx, byref(ms)))
        return ms.value

    def _dt(self, dtype):
        return self.dtype if dtype is None else _lib.dtype_code(dtype)

    def _cin(self, x, dt):
        """complex input -> (DeviceArray, was_host)"""
        if isinstance(x, DeviceArray):
            if x.dtype != np.dtype(_lib.np_complex(dt)):
                raise TypeError("device array is %s but the call runs in %s" % (x.dtype, _lib.np_complex(dt)))
            return x, False
        return self.to_device(np.asarray(x), _lib.np_complex(dt)), True

    def _iin(self, x):
        if isinstance(x, DeviceArray):
            if x.dtype != np.dtype(np.int32):
                raise TypeError("index device arrays must be int32")
            return x, False
        return self.to_device(np.asarray(x), np.int32), True

    @staticmethod
    def _out(arr, host, as_dtype=None):
        if not host:
            return arr
        res = arr.get()
        return res.astype(as_dtype) if as_dtype is not None else res

    # ---- constellation ----------------------------------------------------------------------
    def set_constellation(self, symbols, kind=CONST_GENERIC):
        symbols = np.ascontiguousarray(symbols, dtype=np.complex128).reshape(-1)
        key = (symbols.tobytes(), kind)
        if key == self._table_key:
            return
        view = symbols.view(np.float64)
        check(self.lib.mcle_set_constellation(self.ctx, view.ctypes.data_as(ctypes.POINTER(c_double)),
                                              symbols.size, int(kind)))
        self._table_key, self.M = key, symbols.size

    # ---- a2/a3/a4 ---------------------------------------------------------------------------
    def modulate(self, idx, dtype=None):
        dt = self._dt(dtype)
        d_idx, host = self._iin(idx)
        out = self.empty(d_idx.shape, _lib.np_complex(dt))
        rc = self.lib.mcle_modulate(self.ctx, dt, d_idx.ptr, out.ptr, d_idx.size)
        if rc:
            msg = self.lib.mcle_last_error().decode()
            if "between 0 and 2^M" in msg:
                raise ValueError(msg)
            raise McleError(msg)
        return self._out(out, host)

    def demodulate(self, rx, method=DEMOD_MINDIST, dtype=None):
        dt = self._dt(dtype)
        d_rx, host = self._cin(rx, dt)
        out = self.empty(d_rx.shape, np.int32)
        check(self.lib.mcle_demodulate(self.ctx, dt, int(method), d_rx.ptr, out.ptr, d_rx.size))
        return self._out(out, host, np.int64)

    def count_errors(self, tx_idx, rx_idx, bits_per_symbol, n_real=1, counters=None):
        """-> (counters dict, per-realization symbol errors, per-realization bit errors); with a device-resident
        `counters` block (new_counters()) the sums are ADDED there and nothing is read back."""
        a, _ = self._iin(tx_idx)
        b, _ = self._iin(rx_idx)
        if a.size != b.size or a.size % n_real:
            raise ValueError("size mismatch")
        if counters is not None:
            check(self.lib.mcle_count_errors(self.ctx, a.ptr, b.ptr, a.size // n_real, n_real, int(bits_per_symbol),
                                             counters.ptr, None, None))
            return None
        cnt = self.zeros(1, np.dtype((np.void, ctypes.sizeof(Counters))))
        se, be = self.empty(n_real, np.uint32), self.empty(n_real, np.uint32)
        check(self.lib.mcle_count_errors(self.ctx, a.ptr, b.ptr, a.size // n_real, n_real, int(bits_per_symbol),
                                         cnt.ptr, se.ptr, be.ptr))
        return self._counters(cnt), se.get(), be.get()

    def demod_count(self, rx, tx_idx, n_real=1, method=DEMOD_MINDIST, dtype=None, counters=None):
        dt = self._dt(dtype)
        d_rx, _ = self._cin(rx, dt)
        if isinstance(tx_idx, DeviceArray) and tx_idx.dtype == np.dtype(np.uint8):     # byte labels (rand_modulate_batch)
            a, fn = tx_idx, self.lib.mcle_demod_count_u8
        else:
            a, _ = self._iin(tx_idx)
            fn = self.lib.mcle_demod_count
        if a.size != d_rx.size or a.size % n_real:
            raise ValueError("size mismatch")
        if counters is not None:
            check(fn(self.ctx, dt, int(method), d_rx.ptr, a.ptr, a.size // n_real, n_real, counters.ptr, None, None))
            return None
        cnt = self.zeros(1, np.dtype((np.void, ctypes.sizeof(Counters))))
        se, be = self.empty(n_real, np.uint32), self.empty(n_real, np.uint32)
        check(fn(self.ctx, dt, int(method), d_rx.ptr, a.ptr, a.size // n_real, n_real, cnt.ptr, se.ptr, be.ptr))
        return self._counters(cnt), se.get(), be.get()

    def _counters(self, cnt):
        raw = cnt.get().tobytes()
        return Counters.from_buffer_copy(raw).as_dict()

    # ---- a5 ---------------------------------------------------------------------------------
    def randn_c(self, n, seed, realization, stream=_lib.STREAM_NOISE, first=0, variance=1.0, dtype=None,
                device=False):
        dt = self._dt(dtype)
        out = self.empty(n, _lib.np_complex(dt))
        check(self.lib.mcle_randn_c(self.ctx, dt, int(seed), int(realization), int(stream), int(first),
                                    float(variance), out.ptr, out.size))
        return out if device else out.get()

    def rand_symbols(self, n, M, seed, realization, first=0, device=False):
        out = self.empty(n, np.int32)
        check(self.lib.mcle_rand_symbols(self.ctx, int(seed), int(realization), int(first), int(M), out.ptr,
                                         out.size))
        return out if device else out.get().astype(np.int64)

    def awgn_add(self, x, noise, noise_var, dtype=None):
        dt = self._dt(dtype)
        d_x, host = self._cin(x, dt)
        d_n, _ = self._cin(noise, dt)
        if d_x.size != d_n.size:
            raise ValueError("size mismatch")
        out = self.empty(d_x.shape, _lib.np_complex(dt))
        check(self.lib.mcle_awgn_add(self.ctx, dt, d_x.ptr, d_n.ptr, float(noise_var), out.ptr, d_x.size))
        return self._out(out, host)

    def cmul(self, a, b, dtype=None):
        dt = self._dt(dtype)
        d_a, host = self._cin(a, dt)
        d_b, _ = self._cin(b, dt)
        if d_a.size != d_b.size:
            raise ValueError("size mismatch")
        out = self.empty(d_a.shape, _lib.np_complex(dt))
        check(self.lib.mcle_cmul(self.ctx, dt, d_a.ptr, d_b.ptr, out.ptr, d_a.size))
        return self._out(out, host)

    def cdiv(self, num, den, dtype=None):
        dt = self._dt(dtype)
        a, host = self._cin(num, dt)
        b, _ = self._cin(den, dt)
        if a.size != b.size:
            raise ValueError("size mismatch")
        out = self.empty(a.shape, _lib.np_complex(dt))
        check(self.lib.mcle_cdiv(self.ctx, dt, a.ptr, b.ptr, out.ptr, a.size))
        return self._out(out, host)

    # ---- a6/a9 ------------------------------------------------------------------------------
    def jakes_generate(self, phi, psi, Fd, t0, dt_step, n_samples, tap_power=None, dtype=None, device=False,
                       times=None):
        """phi, psi: [L, n_streams]; returns h [n_streams, n_samples] at t0 + k*dt_step, or at the explicit
        sample times `times` (float64 [n_samples])."""
        dt = self._dt(dtype)
        phi = np.ascontiguousarray(phi, dtype=np.float64)
        psi = np.ascontiguousarray(psi, dtype=np.float64)
        L, S = phi.shape
        out = self.empty((S, n_samples), _lib.np_complex(dt))
        pw = None
        if tap_power is not None:
            pw_arr = np.ascontiguousarray(tap_power, dtype=np.float64)
            pw = pw_arr.ctypes.data_as(ctypes.POINTER(c_double))
        dp = ctypes.POINTER(c_double)
        if times is not None:
            tt = np.ascontiguousarray(times, dtype=np.float64)
            if tt.size != n_samples:
                raise ValueError("times must have n_samples entries")
            check(self.lib.mcle_jakes_generate_at(self.ctx, dt, phi.ctypes.data_as(dp), psi.ctypes.data_as(dp), L, S,
                                                  float(Fd), tt.ctypes.data_as(dp), pw, out.ptr, int(n_samples)))
        else:
            check(self.lib.mcle_jakes_generate(self.ctx, dt, phi.ctypes.data_as(dp), psi.ctypes.data_as(dp), L, S,
                                               float(Fd), float(t0), float(dt_step), pw, out.ptr, int(n_samples)))
        return out if device else out.get()

    def tdl_apply(self, x, taps, delays, dtype=None):
        dt = self._dt(dtype)
        d_x, host = self._cin(x, dt)
        d_g, _ = self._cin(taps, dt)
        delays = np.ascontiguousarray(delays, dtype=np.int32)
        n = d_x.size
        if d_g.size != n * delays.size:
            raise ValueError("taps must be [n_taps, n]")
        out = self.empty(n + int(delays[-1]), _lib.np_complex(dt))
        check(self.lib.mcle_tdl_apply(self.ctx, dt, d_x.ptr, d_g.ptr, delays.ctypes.data_as(ctypes.POINTER(c_int32)),
                                      delays.size, out.ptr, n))
        return self._out(out, host)

    def tdl_apply_mimo(self, x, taps, delays, dtype=None):
        """x [nt, n], taps [S, nr, nt, n] -> y [nr, n + max delay] (fading.py:1107-1117)."""
        dt = self._dt(dtype)
        d_x, host = self._cin(x, dt)
        d_g, _ = self._cin(taps, dt)
        delays = np.ascontiguousarray(delays, dtype=np.int32)
        batched = len(d_g.shape) == 5            # [batch, S, nr, nt, n] with x [batch, nt, n]
        b = d_g.shape[0] if batched else 1
        S, nr, nt, n = d_g.shape[-4:]
        if tuple(d_x.shape[-2:]) != (nt, n) or S != delays.size or d_x.size != b * nt * n:
            raise ValueError("x must be [nt, n] and taps [n_taps, nr, nt, n]")
        shape = (nr, n + int(delays.max()))
        out = self.empty(((b,) + shape) if batched else shape, _lib.np_complex(dt))
        self._raise_value(self.lib.mcle_tdl_apply_mimo(self.ctx, dt, d_x.ptr, d_g.ptr,
                                                       delays.ctypes.data_as(ctypes.POINTER(c_int32)), S, nr, nt,
                                                       out.ptr, n, b))
        return self._out(out, host)

    def tdl_mean_freq_response(self, taps, delays, n_sym, fft_size, cp_size, num_used, dtype=None, group=None,
                               batch=None):
        """taps [S, *links, n_sym*(fft+cp)] -> H [n_sym, num_used, *links] on the used subcarriers; with
        ``group=g`` instead: taps [S, *links, n_sym*g] -> H [n_sym, fft_size, *links], all bins in natural
        order, each averaged over g consecutive samples (g = 1: per-sample response)."""
        dt = self._dt(dtype)
        d_g, host = self._cin(taps, dt)
        delays = np.ascontiguousarray(delays, dtype=np.int32)
        links = d_g.shape[1:-1] if batch is None else d_g.shape[2:-1]   # batch: taps [batch, S, *links, n]
        P = int(np.prod(links)) if links else 1
        if group is not None:
            num_used, cp_size = -int(group), 0
        shape = (n_sym, fft_size if group is not None else num_used) + tuple(links)
        out = self.empty(shape if batch is None else (batch,) + shape, _lib.np_complex(dt))
        self._raise_value(self.lib.mcle_tdl_mean_freq_response(
            self.ctx, dt, d_g.ptr, delays.ctypes.data_as(ctypes.POINTER(c_int32)), delays.size, P, n_sym, fft_size,
            cp_size, num_used, out.ptr, 1 if batch is None else int(batch)))
        return self._out(out, host)

    def jakes_taps_philox(self, seed, first, count, L, Fd, t0, dt_step, stream_amp, n_samples, dtype=None):
        """Device-resident Jakes taps [count, n_streams, n_samples] with the phases of realization first + r
        drawn on-chip (PHASE stream); stream_amp[s] = sqrt(tap power / L)."""
        dt = self._dt(dtype)
        amp = np.ascontiguousarray(stream_amp, dtype=np.float64)
        out = self.empty((count, amp.size, n_samples), _lib.np_complex(dt))
        self._raise_value(self.lib.mcle_jakes_taps_philox(
            self.ctx, dt, int(seed), int(first), int(count), int(L), amp.size, float(Fd), float(t0), float(dt_step),
            amp.ctypes.data_as(ctypes.POINTER(c_double)), out.ptr, int(n_samples)))
        return out

    def awgn_philox(self, x, seed, first, count, noise_var, dtype=None):
        """x [count, ...] (device or host) + sqrt(noise_var) * CN(0,1) of (seed, first + r, NOISE), row-major."""
        dt = self._dt(dtype)
        d_x, host = self._cin(x, dt)
        out = self.empty(d_x.shape, _lib.np_complex(dt))
        self._raise_value(self.lib.mcle_awgn_philox(self.ctx, dt, d_x.ptr, int(seed), int(first), int(count),
                                                    d_x.size // max(int(count), 1), float(noise_var), out.ptr))
        return self._out(out, host)

    def rand_symbols_batch(self, n, M, seed, first, count):
        """Device int32 [count, n]: symbol i of realization first + r."""
        out = self.empty((count, n), np.int32)
        self._raise_value(self.lib.mcle_rand_symbols_batch(self.ctx, int(seed), int(first), int(count), int(M),
                                                           out.ptr, int(n)))
        return out

    def rand_modulate_batch(self, n, seed, first, count, dtype=None, labels=np.int32):
        """(labels [count, n], samples [count, n]) of realizations first .. first + count - 1 for the bound
        constellation: rand_symbols_batch and modulate in one pass over HBM.  labels: np.int32 (default) or np.uint8
        (byte labels, M <= 256: a quarter of the label traffic; demod_count takes either)."""
        dt = self._dt(dtype)
        labels = np.dtype(labels)
        if labels not in (np.dtype(np.int32), np.dtype(np.uint8)):
            raise ValueError("labels must be int32 or uint8")
        idx = self.empty((count, n), labels)
        sym = self.empty((count, n), _lib.np_complex(dt))
        fn = self.lib.mcle_rand_modulate_batch_u8 if labels == np.dtype(np.uint8) else self.lib.mcle_rand_modulate_batch
        self._raise_value(fn(self.ctx, dt, int(seed), int(first), int(count), idx.ptr, sym.ptr, int(n)))
        return idx, sym

    def slice_rows(self, x, row_len):
        """Device copy of x[..., :row_len] (x contiguous [..., n])."""
        n = x.shape[-1]
        rows = x.size // n
        out = self.empty(tuple(x.shape[:-1]) + (row_len,), x.dtype)
        isz = np.dtype(x.dtype).itemsize
        check(self.lib.mcle_memcpy_2d(self.ctx, out.ptr, row_len * isz, x.ptr, n * isz, row_len * isz, rows))
        return out

    def blast_decode_per_subcarrier(self, G, Y, dtype=None):
        """G [ns, nt, nr], Y [nr, ns] -> est [ns*nt] with est[c*nt + a]; or batched G [b, ns, nt, nr],
        Y [b, nr, ns] -> est [b, ns*nt]."""
        dt = self._dt(dtype)
        d_G, _ = self._cin(G, dt)
        d_Y, host = self._cin(Y, dt)
        batched = len(d_G.shape) == 4
        ns, nt, nr = d_G.shape[-3:]
        b = d_G.shape[0] if batched else 1
        out = self.empty((b, ns * nt) if batched else ns * nt, _lib.np_complex(dt))
        self._raise_value(self.lib.mcle_blast_decode_per_subcarrier(self.ctx, dt, d_G.ptr, d_Y.ptr, nr, nt, ns,
                                                                    out.ptr, b))
        return self._out(out, host)

    # ---- a10/a11 ----------------------------------------------------------------------------
    def ofdm_modulate(self, x, fft_size, cp_size, num_used, batch=1, dtype=None):
        dt = self._dt(dtype)
        d_x, host = self._cin(x, dt)
        n_in = d_x.size // batch
        n_sym = -(-n_in // num_used)
        out = self.empty((batch, n_sym * (fft_size + cp_size)), _lib.np_complex(dt))
        rc = self.lib.mcle_ofdm_modulate(self.ctx, dt, d_x.ptr, n_in, fft_size, cp_size, num_used, out.ptr, batch)
        self._raise_value(rc)
        return self._out(out, host)

    def ofdm_demodulate(self, y, fft_size, cp_size, num_used, batch=1, dtype=None):
        dt = self._dt(dtype)
        d_y, host = self._cin(y, dt)
        n_sym = (d_y.size // batch) // (fft_size + cp_size)
        out = self.empty((batch, n_sym * num_used), _lib.np_complex(dt))
        rc = self.lib.mcle_ofdm_demodulate(self.ctx, dt, d_y.ptr, n_sym, fft_size, cp_size, num_used, out.ptr, batch)
        self._raise_value(rc)
        return self._out(out, host)

    def onetap_equalize(self, data, taps, delays, fft_size, cp_size, num_used, dtype=None):
        dt = self._dt(dtype)
        d_d, host = self._cin(data, dt)
        d_g, _ = self._cin(taps, dt)
        delays = np.ascontiguousarray(delays, dtype=np.int32)
        n_sym = d_d.size // num_used
        if d_g.size != delays.size * n_sym * (fft_size + cp_size):
            raise ValueError("taps must be [n_taps, n_sym*(fft+cp)]")
        out = self.empty(d_d.size, _lib.np_complex(dt))
        check(self.lib.mcle_onetap_equalize(self.ctx, dt, d_d.ptr, d_g.ptr,
                                            delays.ctypes.data_as(ctypes.POINTER(c_int32)), delays.size, n_sym,
                                            fft_size, cp_size, num_used, out.ptr))
        return self._out(out, host)

    def _raise_value(self, rc):
        if rc:
            msg = self.lib.mcle_last_error().decode()
            if rc == -1:
                raise ValueError(msg)
            raise McleError(msg)

    # ---- a12 --------------------------------------------------------------------------------
    def blast_encode(self, x, nt, batch=1, dtype=None):
        dt = self._dt(dtype)
        d_x, host = self._cin(x, dt)
        n = d_x.size // batch
        out = self.empty((batch, nt, n // nt if nt else 0), _lib.np_complex(dt))
        self._raise_value(self.lib.mcle_blast_encode(self.ctx, dt, d_x.ptr, nt, n, out.ptr, batch))
        return self._out(out, host)

    def blast_filter(self, H, noise_var, dtype=None, read_skipped=True):
        """H [batch, nr, nt] -> (G [batch, nt, nr], skipped [batch]); read_skipped=False leaves the flags on the
        device (returned as a DeviceArray) and saves the blocking copy."""
        dt = self._dt(dtype)
        d_H, host = self._cin(H, dt)
        b, nr, nt = d_H.shape
        G = self.empty((b, nt, nr), _lib.np_complex(dt))
        sk = self.empty(b, np.uint32)
        self._raise_value(self.lib.mcle_blast_filter(self.ctx, dt, d_H.ptr, nr, nt, float(noise_var), G.ptr, sk.ptr,
                                                     b))
        return self._out(G, host), (sk.get() if read_skipped else sk)

    def blast_decode(self, G, Y, dtype=None):
        """G [batch, nt, nr], Y [batch, nr, ns] -> est [batch, nt*ns] (Fortran interleave)."""
        dt = self._dt(dtype)
        d_G, _ = self._cin(G, dt)
        d_Y, host = self._cin(Y, dt)
        b, nt, nr = d_G.shape
        ns = d_Y.shape[-1]
        out = self.empty((b, nt * ns), _lib.np_complex(dt))
        self._raise_value(self.lib.mcle_blast_decode(self.ctx, dt, d_G.ptr, d_Y.ptr, nr, nt, ns, out.ptr, b))
        return self._out(out, host)

    def mimo_channel(self, H, X, noise=None, noise_var=0.0, dtype=None):
        dt = self._dt(dtype)
        d_H, _ = self._cin(H, dt)
        d_X, host = self._cin(X, dt)
        b, nr, nt = d_H.shape
        ns = d_X.shape[-1]
        d_n = None
        if noise is not None:
            d_n, _ = self._cin(noise, dt)
        out = self.empty((b, nr, ns), _lib.np_complex(dt))
        self._raise_value(self.lib.mcle_mimo_channel(self.ctx, dt, d_H.ptr, d_X.ptr, d_n.ptr if d_n else None,
                                                     float(noise_var), nr, nt, ns, out.ptr, b))
        return self._out(out, host)

    def mimo_channel_philox(self, H, X, seed, first, noise_var, dtype=None):
        """Y[b] = H[b] X[b] + sqrt(noise_var) * CN(0,1) drawn on-chip (NOISE stream of realization first + b, sample
        r*ns + c): the channel operator of the staged config-4 chain in one pass."""
        dt = self._dt(dtype)
        d_H, _ = self._cin(H, dt)
        d_X, host = self._cin(X, dt)
        b, nr, nt = d_H.shape
        ns = d_X.shape[-1]
        out = self.empty((b, nr, ns), _lib.np_complex(dt))
        self._raise_value(self.lib.mcle_mimo_channel_philox(self.ctx, dt, d_H.ptr, d_X.ptr, int(seed), int(first),
                                                            float(noise_var), nr, nt, ns, out.ptr, b))
        return self._out(out, host)

    def randn_c_batch(self, n, seed, first, count, stream=_lib.STREAM_CHAN, variance=1.0, dtype=None):
        """Device complex [count, n]: CN(0, variance) sample i of (seed, first + r, stream)."""
        dt = self._dt(dtype)
        out = self.empty((count, n), _lib.np_complex(dt))
        self._raise_value(self.lib.mcle_randn_c_batch(self.ctx, dt, int(seed), int(first), int(count), int(stream),
                                                      int(n), float(variance), out.ptr))
        return out

    # ---- a13: Alamouti / MRT / SVD ----------------------------------------------------------
    def alamouti_encode(self, x, batch=1, dtype=None):
        dt = self._dt(dtype)
        d_x, host = self._cin(x, dt)
        n = d_x.size // batch
        out = self.empty((batch, 2, n), _lib.np_complex(dt))
        self._raise_value(self.lib.mcle_alamouti_encode(self.ctx, dt, d_x.ptr, n, out.ptr, batch))
        return self._out(out, host)

    def alamouti_decode(self, H, Y, dtype=None):
        """H [batch, nr, 2], Y [batch, nr, n] -> [batch, n]."""
        dt = self._dt(dtype)
        d_H, _ = self._cin(H, dt)
        d_Y, host = self._cin(Y, dt)
        b, nr, _two = d_H.shape
        n = d_Y.shape[-1]
        out = self.empty((b, n), _lib.np_complex(dt))
        self._raise_value(self.lib.mcle_alamouti_decode(self.ctx, dt, d_H.ptr, d_Y.ptr, nr, n, out.ptr, b))
        return self._out(out, host)

    def mrt_encode(self, h, x, dtype=None):
        """h [batch, nt], x [batch, n] -> [batch, nt, n]."""
        dt = self._dt(dtype)
        d_h, _ = self._cin(h, dt)
        d_x, host = self._cin(x, dt)
        b, nt = d_h.shape
        n = d_x.size // b
        out = self.empty((b, nt, n), _lib.np_complex(dt))
        self._raise_value(self.lib.mcle_mrt_encode(self.ctx, dt, d_h.ptr, d_x.ptr, nt, n, out.ptr, b))
        return self._out(out, host)

    def mrt_decode(self, h, y, dtype=None):
        dt = self._dt(dtype)
        d_h, _ = self._cin(h, dt)
        d_y, host = self._cin(y, dt)
        b, nt = d_h.shape
        n = d_y.size // b
        out = self.empty((b, n), _lib.np_complex(dt))
        self._raise_value(self.lib.mcle_mrt_decode(self.ctx, dt, d_h.ptr, d_y.ptr, nt, n, out.ptr, b))
        return self._out(out, host)

    def svd_filters(self, H, dtype=None):
        """H [batch, n, n] -> (W [batch, n, n] precoder, G [batch, n, n] receive filter, S [batch, n])."""
        dt = self._dt(dtype)
        d_H, host = self._cin(H, dt)
        b, n, _n = d_H.shape
        W, G = self.empty((b, n, n), _lib.np_complex(dt)), self.empty((b, n, n), _lib.np_complex(dt))
        S = self.empty((b, n), np.float64)
        self._raise_value(self.lib.mcle_svd_filters(self.ctx, dt, d_H.ptr, n, W.ptr, G.ptr, S.ptr, b))
        return self._out(W, host), self._out(G, host), S.get()

    def gmd_filters(self, H, noise_var=0.0, dtype=None):
        """H [batch, n, n] -> (W precoder, G receive filter, R [batch, n, n] real upper triangular)."""
        dt = self._dt(dtype)
        d_H, host = self._cin(H, dt)
        b, n, _n = d_H.shape
        W, G = self.empty((b, n, n), _lib.np_complex(dt)), self.empty((b, n, n), _lib.np_complex(dt))
        R, sk = self.empty((b, n, n), np.float64), self.empty(b, np.uint32)
        self._raise_value(self.lib.mcle_gmd_filters(self.ctx, dt, d_H.ptr, n, float(noise_var), W.ptr, G.ptr, R.ptr,
                                                    sk.ptr, b))
        return self._out(W, host), self._out(G, host), R.get()

    def post_processing_sinrs(self, H, W, G_H, noise_var):
        """calc_post_processing_linear_SINRs (mimo.py:62-118), batched: H [b, nr, nt], W [b, nt, ns],
        G_H [b, ns, nr] (complex128) -> linear SINRs [b, ns]."""
        H = np.ascontiguousarray(H, dtype=np.complex128)
        W = np.ascontiguousarray(W, dtype=np.complex128)
        G = np.ascontiguousarray(G_H, dtype=np.complex128)
        b, nr, nt = H.shape
        ns = W.shape[2]
        if W.shape != (b, nt, ns) or G.shape != (b, ns, nr):
            raise ValueError("shapes must be H [b, nr, nt], W [b, nt, ns], G_H [b, ns, nr]")
        out = self.empty((b, ns), np.float64)
        dH, dW, dG = self.to_device(H), self.to_device(W), self.to_device(G)
        self._raise_value(self.lib.mcle_post_processing_sinrs(self.ctx, dH.ptr, dW.ptr, dG.ptr, float(noise_var or 0.0),
                                                              nr, nt, ns, out.ptr, b))
        return out.get()

    def mu_link_stats(self, big_H, Nr, Nt, F=None, U=None, noise_var=0.0, pe=1.0, n_ext=0, joint=False, pathloss_big=None,
                      want=("Q", "Re", "B", "sinr")):
        """MultiUserChannelMatrix(ExtInt).calc_Q / calc_JP_Q / _calc_Bkl_cov_matrix_all_l / calc_SINR / calc_JP_SINR /
        calc_cov_matrix_extint_plus_noise (channels/multiuser.py:1314-2008, :2469-2807) for a batch of channels.
        big_H [b, sum Nr, sum Nt + n_ext]; F: per user [b?, rows, Ns_k] (rows = Nt[k], or sum Nt when `joint`); U: per
        user [b?, Nr[k], Ns_k].  -> dict of per-user lists: Q[k] [b, Nr_k, Nr_k], Re[k], B[k] [b, Ns_k, Nr_k, Nr_k],
        sinr[k] [b, Ns_k]."""
        from ._lib import MuStatsCfg
        H = np.ascontiguousarray(big_H, dtype=np.complex128)
        if H.ndim == 2:
            H = H[None]
        Nr, Nt = [int(v) for v in Nr], [int(v) for v in Nt]
        K, b = len(Nr), H.shape[0]
        if K < 1 or K > 4 or len(Nt) != K or max(Nr + Nt) > 4 or min(Nr + Nt) < 1 or not 0 <= int(n_ext) <= 8:
            raise ValueError("mu_link_stats covers up to 4 users with up to 4 antennas per node and 8 external antennas")
        if H.shape[1:] != (sum(Nr), sum(Nt) + int(n_ext)):
            raise ValueError("big_H must be [sum(Nr), sum(Nt) + n_ext]")
        cfg = MuStatsCfg()
        cfg.K, cfg.n_ext, cfg.joint, cfg.noise_var, cfg.pe = K, int(n_ext), 1 if joint else 0, float(noise_var or 0.0), float(pe)
        ns = [0] * K
        Fp = np.zeros((b, K, 16, 4), dtype=np.complex128)
        Up = np.zeros((b, K, 4, 4), dtype=np.complex128)
        if F is not None:
            for k in range(K):
                Fk = np.asarray(F[k], dtype=np.complex128)
                Fk = Fk.reshape(Fk.shape[0], -1) if Fk.ndim < 3 else Fk
                rows = sum(Nt) if joint else Nt[k]
                if Fk.shape[-2] != rows or Fk.shape[-1] > 4:
                    raise ValueError("precoder of user %d must have %d rows and at most 4 streams" % (k, rows))
                ns[k] = Fk.shape[-1]
                Fp[:, k, :rows, :ns[k]] = Fk
        if U is not None:
            for k in range(K):
                Uk = np.asarray(U[k], dtype=np.complex128)
                Uk = Uk.reshape(Uk.shape[0], -1) if Uk.ndim < 3 else Uk
                if Uk.shape[-2] != Nr[k] or Uk.shape[-1] != ns[k]:
                    raise ValueError("receive filter of user %d must be [Nr, Ns] = [%d, %d]" % (k, Nr[k], ns[k]))
                Up[:, k, :Nr[k], :ns[k]] = Uk
        for k in range(K):
            cfg.nr[k], cfg.nt[k], cfg.ns[k] = Nr[k], Nt[k], ns[k]
        d_pl = None
        if pathloss_big is not None:
            pl = np.ascontiguousarray(pathloss_big, dtype=np.float64)
            if pl.shape != H.shape[1:]:
                raise ValueError("the path loss matrix must have big_H's shape")
            d_pl = self.to_device(pl)
        dQ = self.empty((b, K, 4, 4), np.complex128) if "Q" in want else None
        dRe = self.empty((b, K, 4, 4), np.complex128) if "Re" in want else None
        dB = self.empty((b, K, 4, 4, 4), np.complex128) if "B" in want else None
        dS = self.empty((b, K, 4), np.float64) if ("sinr" in want and U is not None) else None
        dH, dF = self.to_device(H), self.to_device(Fp)
        dU = self.to_device(Up) if U is not None else None
        self._raise_value(self.lib.mcle_mu_link_stats(self.ctx, byref(cfg), dH.ptr, d_pl.ptr if d_pl else None, dF.ptr,
                                                      dU.ptr if dU else None, dQ.ptr if dQ else None,
                                                      dRe.ptr if dRe else None, dB.ptr if dB else None,
                                                      dS.ptr if dS else None, b))
        out = {"ns": ns}
        if dQ:
            q = dQ.get()
            out["Q"] = [q[:, k, :Nr[k], :Nr[k]] for k in range(K)]
        if dRe:
            r = dRe.get()
            out["Re"] = [r[:, k, :Nr[k], :Nr[k]] for k in range(K)]
        if dB:
            bb = dB.get()
            out["B"] = [bb[:, k, :ns[k], :Nr[k], :Nr[k]] for k in range(K)]
        if dS:
            sv = dS.get()
            out["sinr"] = [sv[:, k, :ns[k]] for k in range(K)]
        return out

    # ---- fused pipelines --------------------------------------------------------------------
    def _run(self, fn, cfg, seed, first, count, dtype, per_realization, counters=None):
        dt = self._dt(dtype)
        cnt = counters if counters is not None else self.zeros(1, np.dtype((np.void, ctypes.sizeof(Counters))))
        se = be = None
        if per_realization:
            se, be = self.empty(count, np.uint32), self.empty(count, np.uint32)
        check(fn(self.ctx, dt, byref(cfg), int(seed), int(first), int(count), cnt.ptr,
                 se.ptr if se else None, be.ptr if be else None))
        if counters is not None:
            return None
        res = self._counters(cnt)
        if per_realization:
            return res, se.get(), be.get()
        return res

    # ---- multi-GPU exchange (csrc/comm.hip): one RCCL all-reduce of the counter blocks ----------------
    @staticmethod
    def _torch_before_rccl():
        """Load order matters in a process that has PyTorch installed: initialising RCCL through libmcle and
        importing torch AFTERWARDS ends in 'double free or corruption' at interpreter exit (measured on the box,
        scripts/experiments/rccl_exit_probe.py: torch first is clean, torch later aborts, with torch's bundled
        librccl and with /opt/rocm's alike).  So torch -- if it is there at all -- is imported before the first
        RCCL call; a host without torch (a C caller) has no second runtime to collide with."""
        import importlib.util
        try:
            if importlib.util.find_spec("torch") is not None:
                import torch  # noqa: F401
        except (ImportError, ValueError):
            pass

    def comm_unique_id(self):
        """128-byte RCCL id, drawn by rank 0 and handed to the other ranks by the caller (pyphysim_amd.distributed)."""
        self._torch_before_rccl()
        path = _lib.torch_rccl_path()
        check(self.lib.mcle_comm_load(path.encode() if path else None))
        buf = ctypes.create_string_buffer(128)
        check(self.lib.mcle_comm_unique_id(buf, 128))
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        self._torch_before_rccl()
        path = _lib.torch_rccl_path()
        check(self.lib.mcle_comm_load(path.encode() if path else None))
        check(self.lib.mcle_comm_init(self.ctx, ctypes.c_char_p(bytes(unique_id)), int(rank), int(world)))

    def comm_destroy(self):
        check(self.lib.mcle_comm_destroy(self.ctx))

    def comm_info(self):
        r, w = ctypes.c_int(0), ctypes.c_int(1)
        check(self.lib.mcle_comm_info(self.ctx, byref(r), byref(w)))
        return r.value, w.value

    def counters_allreduce(self, cnt, n=1):
        """In place on the stream: sums of words 0..5, maxima of n_symbols / n_bits, over the ranks."""
        check(self.lib.mcle_counters_allreduce(self.ctx, cnt.ptr, int(n)))

    def allreduce_f64(self, values):
        """Sum of a small float64 vector over the ranks -> NumPy array."""
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        d = self.to_device(v)
        check(self.lib.mcle_allreduce_f64(self.ctx, d.ptr, v.size))
        return d.get()

    def new_counters(self):
        return self.zeros(1, np.dtype((np.void, ctypes.sizeof(Counters))))

    def read_counters(self, cnt):
        return self._counters(cnt)

    def run_awgn(self, n_symbols, noise_var, seed, first, count, method=DEMOD_MINDIST, dtype=None,
                 per_realization=False, counters=None):
        cfg = AwgnCfg(int(n_symbols), int(method), float(noise_var))
        return self._run(self.lib.mcle_run_awgn, cfg, seed, first, count, dtype, per_realization, counters)

    def run_flat_fading(self, n_symbols, noise_var, seed, first, count, Fd=100.0, Ts=1e-3, L=8,
                        rayleigh_iid=False, method=DEMOD_MINDIST, dtype=None, per_realization=False,
                        counters=None):
        cfg = FlatCfg(int(n_symbols), int(method), float(noise_var), float(Fd), float(Ts), int(L),
                      1 if rayleigh_iid else 0)
        return self._run(self.lib.mcle_run_flat_fading, cfg, seed, first, count, dtype, per_realization, counters)

    def run_ofdm_tdl(self, fft_size, cp_size, num_used, n_ofdm_sym, noise_var, tap_power, tap_delay, seed, first,
                     count, Fd=10.0, Ts=1.0 / (15e3 * 1024), L=8, method=DEMOD_MINDIST, dtype=None,
                     per_realization=False, counters=None):
        cfg = OfdmTdlCfg()
        cfg.fft_size, cfg.cp_size, cfg.num_used, cfg.n_ofdm_sym = int(fft_size), int(cp_size), int(num_used), int(
            n_ofdm_sym)
        cfg.demod_method, cfg.n_taps, cfg.L = int(method), len(tap_delay), int(L)
        cfg.noise_var, cfg.Fd, cfg.Ts = float(noise_var), float(Fd), float(Ts)
        if len(tap_delay) > _lib.MAX_TAPS or len(tap_power) != len(tap_delay):
            raise ValueError("at most %d taps; powers and delays must match" % _lib.MAX_TAPS)
        for i, (p, d) in enumerate(zip(tap_power, tap_delay)):
            cfg.tap_power[i], cfg.tap_delay[i] = float(p), int(d)
        return self._run(self.lib.mcle_run_ofdm_tdl, cfg, seed, first, count, dtype, per_realization, counters)

    def run_mimo_ofdm(self, nt, nr, fft_size, cp_size, num_used, n_ofdm_sym, noise_var, seed, first, count,
                      mmse=True, method=DEMOD_MINDIST, dtype=None, per_realization=False, counters=None):
        cfg = MimoOfdmCfg(int(nt), int(nr), int(fft_size), int(cp_size), int(num_used), int(n_ofdm_sym),
                          int(method), 1 if mmse else 0, float(noise_var))
        return self._run(self.lib.mcle_run_mimo_ofdm, cfg, seed, first, count, dtype, per_realization, counters)

    def run_mimo_flat(self, scheme, nt, nr, n_symbols, noise_var, seed, first, count, mmse=False,
                      method=DEMOD_MINDIST, dtype=None, per_realization=False, counters=None):
        """The reference's MIMO application (apps/mimo/simulate_mimo.py:68-142) fused: flat H per realization and
        one of 'blast', 'mrc', 'mrt', 'alamouti', 'svd', 'gmd'; n_symbols per layer."""
        cfg = _lib.MimoFlatCfg(_lib.MIMO_SCHEMES[scheme], int(nt), int(nr), int(n_symbols), int(method),
                               1 if mmse else 0, float(noise_var))
        return self._run(self.lib.mcle_run_mimo_flat, cfg, seed, first, count, dtype, per_realization, counters)

    def run_mimo_ofdm_tdl(self, nt, nr, fft_size, cp_size, num_used, n_ofdm_sym, noise_var, tap_power, tap_delay,
                          seed, first, count, Fd=10.0, Ts=1.0 / (15e3 * 1024), L=8, mmse=True, method=DEMOD_MINDIST,
                          dtype=None, per_realization=False, counters=None):
        """Fused frequency-selective MIMO-OFDM.  Raises _lib.McleUnsupported when the Doppler is
        beyond the kernel's tap model; `simulators.MimoOfdmTdlSimulator` then runs the staged chain."""
        cfg = _lib.MimoOfdmTdlCfg()
        cfg.nt, cfg.nr, cfg.fft_size, cfg.cp_size = int(nt), int(nr), int(fft_size), int(cp_size)
        cfg.num_used, cfg.n_ofdm_sym, cfg.demod_method = int(num_used), int(n_ofdm_sym), int(method)
        cfg.mmse, cfg.n_taps, cfg.L = (1 if mmse else 0), len(tap_delay), int(L)
        cfg.noise_var, cfg.Fd, cfg.Ts = float(noise_var), float(Fd), float(Ts)
        if len(tap_delay) > _lib.MAX_TAPS or len(tap_power) != len(tap_delay):
            raise ValueError("at most %d taps; powers and delays must match" % _lib.MAX_TAPS)
        for i, (p, d) in enumerate(zip(tap_power, tap_delay)):
            cfg.tap_power[i], cfg.tap_delay[i] = float(p), int(d)
        return self._run(self.lib.mcle_run_mimo_ofdm_tdl, cfg, seed, first, count, dtype, per_realization, counters)



    def run_ia(self, n_symbols, noise_var, seed, first, count, method=DEMOD_MINDIST, dtype=None,
               per_realization=False, counters=None, solver="closed_form", max_iterations=50, relative_factor=1e-6,
               initialize_with="random"):
        """Config 5 (K = 3, 2x2, one stream per user) with the closed-form solver or one of the iterative
        ones ('alt_min', 'min_leakage', 'max_sinr'; random initial precoders).  Returns the counter dict with
        the extra keys 'sum_capacity' (+ '_sq') and 'ia_runned_iterations' (+ '_sq') = per-realization values
        summed in index order (host side), and with per_realization=True also (sym_err, bit_err, capacities,
        iterations)."""
        dt = self._dt(dtype)
        cfg = IaCfg(3, 2, 2, 1, int(n_symbols), int(method), float(noise_var), _lib.IA_SOLVERS[solver],
                    int(max_iterations), float(relative_factor), _lib.IA_INITS[initialize_with], 0)
        cnt = counters if counters is not None else self.new_counters()
        se, be = self.empty(count, np.uint32), self.empty(count, np.uint32)
        cap = self.empty(count, np.float64)
        its = self.empty(count, np.uint32)
        check(self.lib.mcle_run_ia(self.ctx, dt, byref(cfg), int(seed), int(first), int(count), cnt.ptr, se.ptr,
                                   be.ptr, cap.ptr, its.ptr))
        if counters is not None:
            return None
        res = self._counters(cnt)
        caps = cap.get()
        iterative = solver != "closed_form"
        iters = its.get().astype(np.int64) if (iterative or per_realization) else None
        if res["n_skipped"] or per_realization:
            sev = se.get()
            valid = sev != 0xFFFFFFFF
            if not valid.all():
                caps_v, iters_v = caps[valid], (iters[valid] if iters is not None else None)
            else:
                caps_v, iters_v = caps, iters
        else:                                   # nothing was skipped: no masks, no per-realization error arrays
            sev, caps_v, iters_v = None, caps, iters
        res["sum_capacity"] = float(caps_v.sum())
        res["sum_capacity_sq"] = float(np.square(caps_v).sum())     # (no BLAS: its thread pool costs more than the sum)
        res["ia_runned_iterations"] = int(iters_v.sum()) if iters_v is not None else 0
        res["ia_runned_iterations_sq"] = int(np.square(iters_v).sum()) if iters_v is not None else 0
        if per_realization:
            return res, sev, be.get(), caps, iters
        return res

    def ia_iterative(self, solver, big_H, F_init, noise_var, max_iterations=50, relative_factor=1e-6,
                     initialize_with="fix"):
        """IterativeIASolverBaseClass.solve (algorithms.py:802-883) with initialize_with='fix': big_H
        [batch, 6, 6], F_init [batch, 3, 2] (unit-norm initial precoders) -> dict(F, U = full_W_H, sinr,
        capacity, iterations, skipped)."""
        H = np.ascontiguousarray(big_H, dtype=np.complex128).reshape(-1, 6, 6)
        F0 = np.ascontiguousarray(F_init, dtype=np.complex128).reshape(-1, 3, 2)
        b = H.shape[0]
        if F0.shape[0] != b:
            raise ValueError("big_H and F_init must have the same batch size")
        d_H, d_F0 = self.to_device(H), self.to_device(F0)
        F, U = self.empty((b, 3, 2), np.complex128), self.empty((b, 3, 2), np.complex128)
        sinr, cap = self.empty((b, 3), np.float64), self.empty(b, np.float64)
        its, sk = self.empty(b, np.uint32), self.empty(b, np.uint32)
        self._raise_value(self.lib.mcle_ia_iterative(
            self.ctx, _lib.IA_SOLVERS[solver], _lib.IA_INITS[initialize_with], d_H.ptr, d_F0.ptr, float(noise_var),
            int(max_iterations),
            float(relative_factor), F.ptr, U.ptr, sinr.ptr, cap.ptr, its.ptr, sk.ptr, b))
        return dict(F=F.get(), U=U.get(), sinr=sinr.get(), capacity=cap.get(), iterations=its.get(),
                    skipped=sk.get())

    def ia_solve_general(self, solver, big_H, K, nr, nt, Ns, noise_var, max_iterations=50, relative_factor=1e-6,
                         F_init=None, select=None):
        """Iterative IA for general geometries (csrc/kernels_ia_general.hip; reference ia/algorithms.py:802-883,
        885-1507, 1853-2260): K <= 4 users with nr x nt <= 6 x 6 antennas, Ns = streams per user (int or list).
        big_H [batch, K nr, K nt].  Padded arrays are D x D with D = 4 when max(nr, nt) <= 4, else 6 (the library's two
        capacities).  F_init: [batch, 4, D, D] (user, nt x ns in the top-left corner; [batch, K, D, D] is padded here)
        -- 'fix' / a captured random start -- or None for the 'svd' start.  select: None, 'greedy' (GreedStreamIASolver)
        or 'brute' (BruteForceStreamIASolver, always from 'svd').  -> dict of padded arrays: F [b, K, D, D] (nt x ns in the
        top-left corner), U = full_W_H [b, K, D, D] (ns x nr), sinr [b, K, D], capacity [b], iterations [b],
        Ns [b, K], skipped [b]."""
        D = 4 if max(int(nr), int(nt)) <= 4 else 6
        H = np.ascontiguousarray(big_H, dtype=np.complex128)
        if H.ndim == 2:
            H = H[np.newaxis]
        b = H.shape[0]
        if H.shape[1:] != (K * nr, K * nt):
            raise ValueError("big_H must be [batch, K*nr, K*nt]")
        cfg = _lib.IaGeneralCfg()
        cfg.K, cfg.nr, cfg.nt = int(K), int(nr), int(nt)
        ns_list = [int(Ns)] * K if np.isscalar(Ns) else [int(n) for n in Ns]
        if len(ns_list) != K:
            raise ValueError("Ns must be an int or one value per user")
        for k in range(4):
            cfg.ns[k] = ns_list[k] if k < K else 0
        cfg.solver = _lib.IA_SOLVERS[solver]
        cfg.initialize_with = 0 if (F_init is not None and select != "brute") else 3
        cfg.max_iterations = int(max_iterations)
        cfg.stream_selection = _lib.IA_STREAM_SELECTION[select]
        cfg.noise_var, cfg.relative_factor = float(noise_var), float(relative_factor)
        d_H = self.to_device(H)
        d_F0 = None
        if cfg.initialize_with == 0:
            F0 = np.ascontiguousarray(F_init, dtype=np.complex128)
            if F0.ndim == 4 and F0.shape[0] == b and F0.shape[1] in (K, 4) and F0.shape[2] <= D and F0.shape[3] <= D:
                pad = np.zeros((b, 4, D, D), dtype=np.complex128)       # any smaller padding is re-padded to the capacity
                pad[:, :F0.shape[1], :F0.shape[2], :F0.shape[3]] = F0
                F0 = pad
            if F0.shape != (b, 4, D, D):
                raise ValueError("F_init must be [batch, 4, %d, %d] (user, nt, ns; zero padded)" % (D, D))
            d_F0 = self.to_device(F0)
        F, U = self.empty((b, 4, D, D), np.complex128), self.empty((b, 4, D, D), np.complex128)
        sinr, cap = self.empty((b, 4, D), np.float64), self.empty(b, np.float64)
        its, ns, sk = self.empty(b, np.uint32), self.empty((b, 4), np.int32), self.empty(b, np.uint32)
        every = self.empty((b, 256), np.float64) if select == "brute" else None
        self._raise_value(self.lib.mcle_ia_solve_general(self.ctx, byref(cfg), d_H.ptr, d_F0.ptr if d_F0 else None,
                                                         F.ptr, U.ptr, sinr.ptr, cap.ptr, its.ptr, ns.ptr, sk.ptr,
                                                         every.ptr if every else None, b))
        out = dict(F=F.get()[:, :K], U=U.get()[:, :K], sinr=sinr.get()[:, :K], capacity=cap.get(),
                   iterations=its.get().astype(np.int64), Ns=ns.get()[:, :K], skipped=sk.get())
        if every:
            out["every_capacity"] = every.get()[:, :int(np.prod(ns_list))]     # one value per stream combination
        return out

    def ia_closed_form(self, big_H, noise_var):
        """big_H [batch, 6, 6] complex128 -> dict(F [batch,3,2], U [batch,3,2], sinr [batch,3],
        capacity [batch], skipped [batch])."""
        H = np.ascontiguousarray(big_H, dtype=np.complex128).reshape(-1, 6, 6)
        b = H.shape[0]
        d_H = self.to_device(H)
        F, U = self.empty((b, 3, 2), np.complex128), self.empty((b, 3, 2), np.complex128)
        sinr, cap, sk = self.empty((b, 3), np.float64), self.empty(b, np.float64), self.empty(b, np.uint32)
        check(self.lib.mcle_ia_closed_form(self.ctx, d_H.ptr, float(noise_var), F.ptr, U.ptr, sinr.ptr, cap.ptr,
                                           sk.ptr, b))
        return dict(F=F.get(), U=U.get(), sinr=sinr.get(), capacity=cap.get(), skipped=sk.get())

    # ---- block diagonalisation (comm/blockdiagonalization.py, comm/waterfilling.py) -------------
    def waterfilling(self, gains, total_power, noise_var=1.0):
        """doWF (waterfilling.py:15-92) on gains [batch, n] (or [n]) -> (powers, water levels)."""
        g = np.ascontiguousarray(gains, dtype=np.float64)
        single = g.ndim == 1
        if single:
            g = g.reshape(1, -1)
        elif g.ndim != 2:
            raise ValueError("gains must be [n] or [batch, n]")
        b, n = g.shape
        d_g = self.to_device(g)
        P, mu = self.empty((b, n), np.float64), self.empty(b, np.float64)
        self._raise_value(self.lib.mcle_waterfilling(self.ctx, d_g.ptr, n, float(total_power), float(noise_var),
                                                     P.ptr, mu.ptr, b))
        P, mu = P.get(), mu.get()
        return (P[0], float(mu[0])) if single else (P, mu)

    def block_diagonalize(self, H, num_users, iPu, noise_var, waterfilling=True):
        """BlockDiagonalizer.block_diagonalize / block_diagonalize_no_waterfilling + calc_receive_filter on
        H [batch, n, n] (n = num_users * antennas per user) -> dict(Ms, newH, W, sigma, skipped)."""
        H = np.ascontiguousarray(H, dtype=np.complex128)
        if H.ndim == 2:
            H = H[None]
        b, n, n2 = H.shape
        if n != n2 or n % int(num_users):
            raise ValueError("block diagonalisation needs a square channel whose size is a multiple of num_users")
        d_H = self.to_device(H)
        Ms, newH, W = (self.empty((b, n, n), np.complex128) for _ in range(3))
        sg, sk = self.empty((b, n), np.float64), self.empty(b, np.uint32)
        self._raise_value(self.lib.mcle_block_diagonalize(self.ctx, d_H.ptr, int(num_users), n // int(num_users),
                                                          float(iPu), float(noise_var), 1 if waterfilling else 0,
                                                          Ms.ptr, newH.ptr, W.ptr, sg.ptr, sk.ptr, b))
        return dict(Ms=Ms.get(), newH=newH.get(), W=W.get(), sigma=sg.get(), skipped=sk.get())

    def bd_extint(self, big_H, num_users, n_ant, n_ext, iPu, noise_var, pe, method="whitening", metric=None,
                  num_streams=0, ns_user=None):
        """Block diagonalisation with external interference (csrc/kernels_bd.hip k_bd_extint; reference
        comm/blockdiagonalization.py:666-1469).  big_H [batch, K r, K r + n_ext] (MultiUserChannelMatrixExtInt.big_H);
        method 'whitening' (WhiteningBD) or 'enhanced' (EnhancedBD with metric None / 'naive' / 'fixed' / 'capacity' /
        'candidates' (report the SINRs of every stream count) / 'per_user' (ns_user given)).
        -> dict(Ms [b, K, K r, r], W [b, K, r, r], Ns [b, K], cand_sinr [b, K, r, r] or None, skipped [b])."""
        H = np.ascontiguousarray(big_H, dtype=np.complex128)
        if H.ndim == 2:
            H = H[np.newaxis]
        b, n = H.shape[0], num_users * n_ant
        if H.shape[1:] != (n, n + n_ext):
            raise ValueError("big_H must be [batch, K*r, K*r + n_ext]")
        cfg = _lib.BdExtIntCfg()
        cfg.num_users, cfg.n_ant_per_user, cfg.n_ext = int(num_users), int(n_ant), int(n_ext)
        cfg.method = {"whitening": 0, "enhanced": 1}[method]
        cfg.metric = _lib.BD_METRICS[metric]
        cfg.num_streams = int(num_streams or 0)
        for k in range(4):
            cfg.ns_user[k] = int(ns_user[k]) if (ns_user is not None and k < len(ns_user)) else 0
        cfg.iPu, cfg.noise_var, cfg.pe = float(iPu), float(noise_var or 0.0), float(pe)
        d_H = self.to_device(H)
        Ms, W = self.empty((b, num_users, n, n_ant), np.complex128), self.empty((b, num_users, n_ant, n_ant), np.complex128)
        ns, sk = self.empty((b, num_users), np.int32), self.empty(b, np.uint32)
        cand = self.empty((b, num_users, n_ant, n_ant), np.float64) if cfg.metric == 4 else None
        self._raise_value(self.lib.mcle_bd_extint(self.ctx, byref(cfg), d_H.ptr, Ms.ptr, W.ptr, ns.ptr,
                                                  cand.ptr if cand is not None else None, sk.ptr, b))
        return dict(Ms=Ms.get(), W=W.get(), Ns=ns.get(), cand_sinr=None if cand is None else cand.get(),
                    skipped=sk.get())

    def pinv(self, A, rcond=1e-15):
        """np.linalg.pinv for small matrices [batch, m, n] (or [m, n]), m, n <= 8."""
        A = np.ascontiguousarray(A, dtype=np.complex128)
        single = A.ndim == 2
        if single:
            A = A[None]
        b, m, n = A.shape
        d_A, out = self.to_device(A), self.empty((b, n, m), np.complex128)
        self._raise_value(self.lib.mcle_pinv(self.ctx, d_A.ptr, m, n, float(rcond), out.ptr, b))
        out = out.get()
        return out[0] if single else out

    def run_bd(self, K, nr, n_symbols, iPu, noise_var, seed, first, count, bd_noise_var=1e-50, pathloss=None,
               waterfilling=True, method=DEMOD_MINDIST, dtype=None, per_realization=False, counters=None):
        """apps/comp_BD/simulate_comp_simple.py:95-140 fused (no external interference): K cells of nr x nr
        antennas, joint block-diagonalising precoder, zero forcing at the users; n_symbols per stream."""
        cfg = _lib.BdCfg()
        cfg.K, cfg.nr, cfg.n_symbols, cfg.demod_method = int(K), int(nr), int(n_symbols), int(method)
        cfg.waterfilling, cfg.has_pathloss = (1 if waterfilling else 0), (0 if pathloss is None else 1)
        cfg.iPu, cfg.noise_var, cfg.bd_noise_var = float(iPu), float(noise_var), float(bd_noise_var)
        if pathloss is not None:
            pl = np.asarray(pathloss, dtype=np.float64)
            if pl.shape != (K, K) or K > 4:
                raise ValueError("pathloss must be a [K, K] matrix (rx user, tx user), K <= 4")
            for i, v in enumerate(pl.reshape(-1)):
                cfg.pathloss[i] = float(v)
        return self._run(self.lib.mcle_run_bd, cfg, seed, first, count, dtype, per_realization, counters)

    # ---- same-seed parity mode (NumPy legacy RandomState on the device) ----------------------
    def legacy_draws(self, program, seed_base, first, count):
        """program: list of ('randint', n, range) / ('randn', n) / ('rand', n).  Realization r gets
        what NumPy draws after np.random.seed(seed_base + first + r).  Returns (ints DeviceArray
        [count, n_int] int32, doubles DeviceArray [count, n_dbl] float64)."""
        kinds = {"randint": 0, "randn": 1, "rand": 2}
        segs = (LegacySeg * len(program))()
        n_int = n_dbl = 0
        for i, item in enumerate(program):
            k = kinds[item[0]]
            segs[i].kind, segs[i].n = k, int(item[1])
            segs[i].range = int(item[2]) if k == 0 else 0
            if k == 0:
                n_int += int(item[1])
            else:
                n_dbl += int(item[1])
        ints = self.empty((count, max(n_int, 1)), np.int32)
        dbls = self.empty((count, max(n_dbl, 1)), np.float64)
        status = self.zeros(count, np.uint32)
        check(self.lib.mcle_legacy_draws(self.ctx, segs, len(program), int(seed_base) & 0xFFFFFFFF, int(first),
                                         int(count), ints.ptr, max(n_int, 1), dbls.ptr, max(n_dbl, 1), status.ptr))
        if status.get().any():
            raise McleError("legacy draw program ran out of words")
        return ints, dbls

    def complex_from_parts(self, re, im, scale=1.0, dtype=None):
        """scale * (re + 1j*im) on the device from two float64 arrays (host or device)."""
        dt = self._dt(dtype)
        d_re = re if isinstance(re, DeviceArray) else self.to_device(np.asarray(re), np.float64)
        d_im = im if isinstance(im, DeviceArray) else self.to_device(np.asarray(im), np.float64)
        out = self.empty(d_re.shape, _lib.np_complex(dt))
        check(self.lib.mcle_complex_from_parts(self.ctx, dt, d_re.ptr, d_im.ptr, float(scale), out.ptr, d_re.size))
        return out


_default = {}


def default_device():
    """Device of this process: torch's current device once torch.distributed is initialised (the launcher set it
    per rank), else LOCAL_RANK of the launcher's environment, else 0 -- so that the mirror classes and simulators
    of every rank compute on that rank's GPU instead of all piling onto GPU 0."""
    import os
    import sys
    torch = sys.modules.get("torch")          # only if the process uses torch already: importing it here would cost every
    if torch is not None:                     # torch-free caller seconds and bind torch's bundled HIP / RCCL libraries
        dist = getattr(torch, "distributed", None)
        try:
            if dist is not None and dist.is_available() and dist.is_initialized() and torch.cuda.is_available():
                return int(torch.cuda.current_device())
        except Exception:
            pass
    for key in ("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID"):
        if key in os.environ:
            try:
                n = _lib.device_count()
            except McleError:
                n = 0
            return int(os.environ[key]) % n if n > 0 else 0
    return 0


def get_engine(device=None):
    """Process-wide default engine per device (created on first use); device None = default_device()."""
    if device is None:
        device = default_device()
    eng = _default.get(device)
    if eng is None:
        eng = _default[device] = Engine(device)
    return eng
