"""ctypes binding of libzkhip.so (include/zkhip.h) plus the host-side mirror of the reference-facing
names for this path: best_multiexp, best_fft, EvaluationDomain, ParamsKZG.commit / commit_lagrange,
evaluate_h  [UPSTREAM-RECALL names from halo2curves 0.4.0 / halo2_proofs, the crates the reference
calls through gen_snark_shplonk at /root/reference/src/helpers.rs:233,299].

There is no CPU fallback: importing works anywhere, but every compute call needs the HIP library and
a GPU and raises ZkhipError otherwise.  PyTorch is used only for device memory and the stream.
"""
import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_rccl_lib_named = False
LIB_PATH = os.environ.get("ZKHIP_LIB") or os.path.join(_DIR, "libzkhip.so")   # ZKHIP_LIB: A/B runs of another build of the same library
_LIB = None


class ZkhipError(RuntimeError):
    pass


class ConstraintSystemFailure(ZkhipError):
    """halo2_proofs plonk::Error::ConstraintSystemFailure (ZKHIP_ECONSTRAINT)."""


ECONSTRAINT = -6


class ZkGraph(C.Structure):
    _fields_ = [("constants", C.c_void_p), ("rotations", C.c_void_p), ("code", C.c_void_p),
                ("n_constants", C.c_uint32), ("n_rotations", C.c_uint32), ("n_code_words", C.c_uint32),
                ("n_calculations", C.c_uint32), ("n_intermediates", C.c_uint32)]


class ZkEvalhArgs(C.Structure):
    _fields_ = [("k", C.c_uint32), ("extended_k", C.c_uint32), ("cs_degree", C.c_uint32), ("blinding_factors", C.c_uint32),
                ("extended_omega", C.c_uint64 * 4), ("g_coset", C.c_uint64 * 4), ("delta", C.c_uint64 * 4),
                ("beta", C.c_uint64 * 4), ("gamma", C.c_uint64 * 4), ("theta", C.c_uint64 * 4), ("y", C.c_uint64 * 4),
                ("n_fixed", C.c_uint32), ("n_advice", C.c_uint32), ("n_instance", C.c_uint32), ("n_challenges", C.c_uint32),
                ("fixed_cosets", C.c_void_p), ("advice_cosets", C.c_void_p), ("instance_cosets", C.c_void_p),
                ("challenges", C.c_void_p),
                ("l0", C.c_void_p), ("l_last", C.c_void_p), ("l_active_row", C.c_void_p),
                ("custom_gates", ZkGraph),
                ("n_perm_columns", C.c_uint32), ("n_perm_sets", C.c_uint32),
                ("perm_column_type", C.c_void_p), ("perm_column_index", C.c_void_p),
                ("perm_sigma_cosets", C.c_void_p), ("perm_product_cosets", C.c_void_p),
                ("n_lookups", C.c_uint32), ("_pad", C.c_uint32),
                ("lookup_graphs", C.c_void_p),
                ("lookup_product_cosets", C.c_void_p), ("lookup_input_cosets", C.c_void_p),
                ("lookup_table_cosets", C.c_void_p)]


# every symbol include/zkhip.h declares (checked by tests/test_abi.py without a GPU)
SYMBOLS = [
    "zkhip_init", "zkhip_destroy", "zkhip_last_error", "zkhip_set_stream", "zkhip_synchronize", "zkhip_set_option", "zkhip_trim", "zkhip_key_release",
    "zkhip_comm_use_library", "zkhip_comm_unique_id", "zkhip_comm_init", "zkhip_comm_init_host", "zkhip_comm_set_host_alltoall", "zkhip_comm_destroy", "zkhip_comm_info", "zkhip_comm_describe", "zkhip_comm_allgather_device", "zkhip_comm_shard_columns",
    "zkhip_comm_trace", "zkhip_comm_trace_read", "zkhip_comm_phase_name",
    "zkhip_kzg_setup_range", "zkhip_srs_load_range", "zkhip_srs_range", "zkhip_malloc", "zkhip_free",
    "zkhip_memcpy_h2d", "zkhip_memcpy_d2h", "zkhip_timer_start", "zkhip_timer_stop_ms",
    "zkhip_profile_enable", "zkhip_profile_select", "zkhip_profile_read", "zkhip_profile_counter",
    "zkhip_srs_load", "zkhip_srs_load_device", "zkhip_srs_free", "zkhip_srs_len", "zkhip_srs_window", "zkhip_kzg_setup", "zkhip_srs_read",
    "zkhip_msm_g1", "zkhip_msm_g1_batch", "zkhip_msm_g1_batch_device", "zkhip_msm_g1_batch_range_device", "zkhip_msm_g1_multi_device", "zkhip_g1_add", "zkhip_g1_to_affine", "zkhip_g1_batch_to_affine",
    "zkhip_g1_to_bytes", "zkhip_commitments_read",
    "zkhip_fft", "zkhip_fft_batch_device",
    "zkhip_domain_new", "zkhip_domain_free", "zkhip_domain_k", "zkhip_domain_extended_k", "zkhip_domain_quotient_poly_degree",
    "zkhip_domain_constants", "zkhip_lagrange_to_coeff_device", "zkhip_coeff_to_lagrange_device",
    "zkhip_coeff_to_extended_device", "zkhip_extended_to_coeff_device", "zkhip_divide_by_vanishing_device",
    "zkhip_domain_cosets", "zkhip_coeff_to_cosets_device", "zkhip_cosets_to_pieces_device", "zkhip_evaluate_h_cosets_device",
    "zkhip_lagrange_to_coeff", "zkhip_coeff_to_extended", "zkhip_extended_to_coeff",
    "zkhip_evaluate_h_device", "zkhip_evaluate_h_rows_device", "zkhip_synth_fill_device", "zkhip_synth_small_device",
    "zkhip_batch_invert_device", "zkhip_eval_polynomial_device", "zkhip_eval_polynomials_at_device", "zkhip_permutation_products_device",
    "zkhip_permute_expression_pair_device", "zkhip_lookup_product_device", "zkhip_grand_products_device",
    "zkhip_linear_combination_device", "zkhip_divide_by_linear_device", "zkhip_kate_division_device", "zkhip_shplonk_open",
    "zkhip_create_proof", "zkhip_create_proof_ex", "zkhip_coset_quotient_applies",
    "zkhip_blake2b_transcript_new", "zkhip_blake2b_transcript_free", "zkhip_blake2b_transcript_callbacks", "zkhip_blake2b_transcript_proof",
    "zkhip_blake2b_transcript_points", "zkhip_blake2b_transcript_challenges",
    "zkhip_evm_transcript_new", "zkhip_evm_transcript_free", "zkhip_evm_transcript_callbacks", "zkhip_evm_transcript_proof",
    "zkhip_evm_transcript_challenges", "zkhip_evm_transcript_points", "zkhip_keccak256",
    "zkhip_poseidon_transcript_new", "zkhip_poseidon_transcript_free", "zkhip_poseidon_transcript_callbacks", "zkhip_poseidon_transcript_proof",
    "zkhip_poseidon_transcript_points", "zkhip_poseidon_transcript_challenges", "zkhip_poseidon_permute", "zkhip_poseidon_permute_plain", "zkhip_poseidon_params",
]


def lib():
    """Loads libzkhip.so.  Fails loudly: there is no other implementation of the path."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ZkhipError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(make -C halo2-zkcert_amd/csrc); there is no CPU fallback")
        # torch bundles its own libamdhip64 (same SONAME).  Load torch first so that libzkhip.so binds
        # to that copy: two HIP runtimes in one process do not share devices or streams.
        import torch  # noqa: F401

        L = C.CDLL(LIB_PATH)
        L.zkhip_last_error.restype = C.c_char_p
        L.zkhip_comm_phase_name.restype = C.c_char_p
        L.zkhip_srs_len.restype = C.c_size_t
        L.zkhip_evm_transcript_new.restype = C.c_void_p
        L.zkhip_evm_transcript_callbacks.restype = C.c_void_p
        L.zkhip_evm_transcript_proof.restype = C.c_size_t
        L.zkhip_evm_transcript_challenges.restype = C.c_size_t
        L.zkhip_blake2b_transcript_new.restype = C.c_void_p
        L.zkhip_blake2b_transcript_callbacks.restype = C.c_void_p
        for f in ("zkhip_blake2b_transcript_proof", "zkhip_blake2b_transcript_points", "zkhip_blake2b_transcript_challenges"):
            getattr(L, f).restype = C.c_size_t
        for f in ("zkhip_domain_k", "zkhip_domain_extended_k", "zkhip_domain_quotient_poly_degree"):
            getattr(L, f).restype = C.c_uint32
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise ZkhipError(f"zkhip error {rc}: {lib().zkhip_last_error().decode()}")


def _p(a):
    return C.c_void_p(a.ctypes.data)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class Context:
    """One per process and GPU.  Kernels run on torch's current stream."""

    def __init__(self, device=0):
        import torch

        self.torch = torch
        if not torch.cuda.is_available():
            raise ZkhipError("no GPU visible: the zkhip path has no CPU fallback")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.h = C.c_void_p()
        _check(lib().zkhip_init(C.byref(self.h), C.c_int(device)))
        self.use_torch_stream()

    def use_torch_stream(self):
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        _check(lib().zkhip_set_stream(self.h, C.c_void_p(s)))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            lib().zkhip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- device memory through torch ----
    def empty(self, n_elems, width=4):
        if os.environ.get("ZKHIP_POISON"):      # test aid (csrc/ctx.hip dev_malloc): output tensors start as garbage, not as whatever the allocator had (often zeros)
            return self.torch.full((n_elems, width), -0x5A5A5A5A5A5A5A5B, dtype=self.torch.int64, device=self.device)
        return self.torch.empty((n_elems, width), dtype=self.torch.int64, device=self.device)

    def to_device(self, arr):
        arr = _u64(arr)
        return self.torch.from_numpy(arr.view(np.int64)).to(self.device)

    def to_host(self, t):
        """device tensor -> host uint64 array; small results (the Fiat-Shamir round trips) go through the library's polled copy"""
        if isinstance(t, np.ndarray):
            return t
        nbytes = t.numel() * t.element_size()
        if nbytes <= (1 << 16) and t.is_contiguous():
            out = np.empty(t.shape, dtype=np.uint64)
            _check(lib().zkhip_memcpy_d2h(self.h, _p(out), C.c_void_p(t.data_ptr()), C.c_size_t(nbytes)))
            return out
        return t.cpu().numpy().view(np.uint64)

    def synchronize(self):
        _check(lib().zkhip_synchronize(self.h))

    def set_option(self, name, value):
        """a tuning knob by its short ("msm_c") or environment ("ZKHIP_MSM_C") name; the environment itself is read once, in zkhip_init"""
        _check(lib().zkhip_set_option(self.h, name.encode(), C.c_int(int(value))))

    def trim(self):
        _check(lib().zkhip_trim(self.h))

    def key_release(self, key_id):
        """frees what the context caches per proving key (sorted lookup tables, the key's columns in the coset layout)"""
        _check(lib().zkhip_key_release(self.h, C.c_uint64(int(key_id))))

    # ---- one proof over several GPUs (zkhip_comm_*): one process per GPU
    world, rank = 1, 0

    def comm_init(self, rank, world, dist=None, transport=None, group=None, src=0):
        """Gives the context its communicator.  transport "rccl" (default): rank 0's ncclUniqueId is broadcast through torch.distributed
        and every rank joins (ncclCommInitRank inside the library; the all-gathers of a proof then run over xGMI).  transport "host"
        (ZKHIP_COMM_TRANSPORT=host, or a torch.distributed backend without device support such as gloo): the same all-gathers staged
        through host memory and torch.distributed — bring-up and tests on a one-GPU box, where RCCL refuses two ranks per device.
        group / src: a communicator over a SUBSET of the job's ranks (bench.py --chain: one leaf proof over a few ranks): `group` is the
        torch.distributed process group of the subset, `rank` / `world` the position in and the size of the subset, `src` the GLOBAL rank of
        its first member (who creates the unique id)."""
        import torch

        if transport is None:
            transport = os.environ.get("ZKHIP_COMM_TRANSPORT") or ("host" if dist is not None and dist.get_backend() == "gloo" else "rccl")
        if transport == "rccl":
            # ZKHIP_RCCL_LIB (read HERE, by the Python binding — the library itself reads no such variable): the collective library the
            # transport binds, through zkhip_comm_use_library; the one-GPU tests name tests/fake_rccl/libfake_rccl.so
            global _rccl_lib_named
            if os.environ.get("ZKHIP_RCCL_LIB") and not _rccl_lib_named:
                _check(lib().zkhip_comm_use_library(os.environ["ZKHIP_RCCL_LIB"].encode()))
                _rccl_lib_named = True
            ids = [None]
            if rank == 0:
                buf = (C.c_uint8 * 128)()
                _check(lib().zkhip_comm_unique_id(buf))
                ids = [bytes(buf)]
            if world > 1:
                dist.broadcast_object_list(ids, src=src, group=group)
            _check(lib().zkhip_comm_init(self.h, (C.c_uint8 * 128)(*ids[0]), C.c_int(rank), C.c_int(world)))
        elif transport == "host":
            def _ag(user, send, recv, nbytes):
                try:
                    mine = torch.frombuffer((C.c_uint8 * nbytes).from_address(send), dtype=torch.uint8)
                    outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
                    if dist.get_backend() == "nccl":
                        dev = [o.to(self.device) for o in outs]
                        dist.all_gather(dev, mine.to(self.device), group=group)
                        outs = [d.cpu() for d in dev]
                    else:
                        dist.all_gather(outs, mine.clone(), group=group)
                    for r, o in enumerate(outs):
                        C.memmove(recv + r * nbytes, o.numpy().ctypes.data, nbytes)
                    return 0
                except Exception as e:   # noqa: BLE001 — never let an exception cross the C boundary
                    print(f"zkhip host all-gather failed: {e}", flush=True)
                    return 1
            self._host_ag = HOST_ALLGATHER_FN(_ag)
            _check(lib().zkhip_comm_init_host(self.h, C.c_int(rank), C.c_int(world), self._host_ag, None))

            def _a2a(user, send, recv, nbytes):     # block r of send -> rank r; block r of recv <- rank r
                try:
                    mine = torch.frombuffer((C.c_uint8 * (nbytes * world)).from_address(send), dtype=torch.uint8).clone()
                    out = torch.empty(nbytes * world, dtype=torch.uint8)
                    if dist.get_backend() == "nccl":
                        o = out.to(self.device)
                        dist.all_to_all_single(o, mine.to(self.device), group=group)
                        out = o.cpu()
                    else:
                        dist.all_to_all_single(out, mine, group=group)
                    C.memmove(recv, out.numpy().ctypes.data, nbytes * world)
                    return 0
                except Exception as e:   # noqa: BLE001
                    print(f"zkhip host all-to-all failed: {e}", flush=True)
                    return 1
            if os.environ.get("ZKHIP_HOST_A2A", "1") != "0":      # 0: let the library emulate it through the all-gather callback
                self._host_a2a = HOST_ALLGATHER_FN(_a2a)
                _check(lib().zkhip_comm_set_host_alltoall(self.h, self._host_a2a, None))
        else:
            raise ValueError(transport)
        self.rank, self.world, self.transport = rank, world, transport

    def comm_init_replay(self, rank, world, library):
        """MEASUREMENT ONLY (bench.py --replay-rank): this process becomes rank `rank` of a `world`-rank communicator whose collective
        library is tools/replay_rccl (zkhip_comm_use_library): no peer exists, every peer contribution is fabricated on the device, so the
        library's RCCL branch issues exactly rank `rank`'s kernels, fences and exchanges on an otherwise idle GPU.  Results computed on such
        a context are wrong by construction."""
        global _rccl_lib_named
        if _rccl_lib_named and _rccl_lib_named != os.path.abspath(library):
            raise ZkhipError("comm_init_replay: another collective library is already bound in this process")
        if not _rccl_lib_named:
            _check(lib().zkhip_comm_use_library(os.fsencode(library)))
            _rccl_lib_named = os.path.abspath(library)
        buf = (C.c_uint8 * 128)()
        _check(lib().zkhip_comm_unique_id(buf))
        _check(lib().zkhip_comm_init(self.h, buf, C.c_int(rank), C.c_int(world)))
        self.rank, self.world, self.transport = rank, world, "rccl"

    shard_points = True      # ParamsKZG.setup / .read on a context with a communicator: point-range shards (True) or whole tables

    def comm_shard(self, mode):
        """how the MSMs of a proof are split over the ranks: "points" — every rank holds 1/N of the window tables and sums its point
        range of every column (large k); "columns" — every rank holds the whole tables and commits columns r, r + N, ... (k <= 19)"""
        if mode not in ("points", "columns"):
            raise ValueError(mode)
        self.shard_points = mode == "points"
        _check(lib().zkhip_comm_shard_columns(self.h, C.c_int(0 if self.shard_points else 1)))

    def comm_destroy(self):
        _check(lib().zkhip_comm_destroy(self.h))
        self.rank, self.world = 0, 1

    def comm_bytes_gathered(self):
        v = C.c_uint64()
        _check(lib().zkhip_comm_info(self.h, None, None, C.byref(v)))
        return v.value

    def comm_describe(self):
        """-> dict(transport, nranks, transport_ranks, bytes_gathered, collectives): what the library's communicator itself reports
        (transport_ranks is ncclCommCount for RCCL: evidence that RCCL joined that many processes)"""
        buf = C.create_string_buffer(96)
        tr, co, by, rk, nr = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_int(), C.c_int()
        _check(lib().zkhip_comm_describe(self.h, buf, C.c_size_t(96), C.byref(tr), C.byref(co)))
        _check(lib().zkhip_comm_info(self.h, C.byref(rk), C.byref(nr), C.byref(by)))
        return dict(transport=buf.value.decode(), rank=rk.value, nranks=nr.value, transport_ranks=tr.value, bytes_gathered=by.value,
                    collectives=co.value)

    def comm_trace(self, on=True):
        """start (clearing the record, marking time zero on the context's stream) / stop the per-exchange trace of the RCCL branch"""
        _check(lib().zkhip_comm_trace(self.h, C.c_int(1 if on else 0)))

    def comm_trace_read(self, cap=4096):
        """-> (entries, end_us): per exchange since the mark dict(phase, bytes_received, host_issue_us, stream_done_us, bulk, kind); end_us = when the
        context's stream drained (zkhip.h)"""
        n, end = C.c_size_t(), C.c_double()
        ph, fl = (C.c_uint8 * cap)(), (C.c_uint8 * cap)()
        by, hu, du = (C.c_uint64 * cap)(), (C.c_double * cap)(), (C.c_double * cap)()
        _check(lib().zkhip_comm_trace_read(self.h, C.c_size_t(cap), C.byref(n), ph, by, hu, du, fl, C.byref(end)))
        m = min(n.value, cap)
        return [dict(phase=lib().zkhip_comm_phase_name(C.c_uint8(ph[i])).decode(), bytes_received=by[i], host_issue_us=round(hu[i], 1), stream_done_us=round(du[i], 1),
                     bulk=bool(fl[i] & 1), kind="sendrecv" if fl[i] & 2 else "allgather") for i in range(m)], end.value

    def comm_allgather(self, send, recv):
        """all-gather of device tensors through the context's communicator (recv: world x send)"""
        _check(lib().zkhip_comm_allgather_device(self.h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()),
                                                 C.c_size_t(send.numel() * send.element_size())))

    def shard_range(self, n):
        """[first, first + count) of n points held by this rank"""
        lo, hi = self.rank * n // self.world, (self.rank + 1) * n // self.world
        return lo, hi - lo

    def timer_start(self):
        _check(lib().zkhip_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_float()
        _check(lib().zkhip_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        _check(lib().zkhip_profile_enable(self.h, C.c_int(1 if on else 0)))

    def profile_select(self, kernel=None):
        """record only this kernel's spans (None = all)"""
        _check(lib().zkhip_profile_select(self.h, kernel.encode() if kernel else None))

    def profile_read(self, kernel):
        """(total_ms, launches) of the named kernel since profile_enable, from HIP events on its stream."""
        ms, n = C.c_double(), C.c_uint64()
        _check(lib().zkhip_profile_read(self.h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_counter(self, name):
        v = C.c_uint64()
        _check(lib().zkhip_profile_counter(self.h, name.encode(), C.byref(v)))
        return v.value

    def synth_fill(self, n, seed, first=0):
        t = self.empty(n)
        _check(lib().zkhip_synth_fill_device(self.h, C.c_void_p(t.data_ptr()), C.c_size_t(n), C.c_uint64(seed), C.c_uint64(first)))
        return t

    def synth_small(self, n, seed, bits_per_mille=900, word_bits=32, first=0):
        """a column of small values: bits with probability bits_per_mille / 1000, else word_bits-bit words (zkhip.h)"""
        t = self.empty(n)
        _check(lib().zkhip_synth_small_device(self.h, C.c_void_p(t.data_ptr()), C.c_size_t(n), C.c_uint64(seed), C.c_uint64(first),
                                              C.c_uint32(bits_per_mille), C.c_uint32(word_bits)))
        return t

    # ---- best_fft ----
    def best_fft(self, a, omega, log_n):
        """halo2curves fft::best_fft(a, omega, log_n) on a host array (returns the transformed copy)."""
        a = _u64(a).copy().reshape(1 << log_n, 4)
        _check(lib().zkhip_fft(self.h, _p(a), _p(_u64(omega)), C.c_uint32(log_n)))
        return a

    def fft_batch_device(self, polys, omega, log_n):
        ptrs = (C.c_void_p * len(polys))(*[p.data_ptr() for p in polys])
        _check(lib().zkhip_fft_batch_device(self.h, ptrs, C.c_size_t(len(polys)), _p(_u64(omega)), C.c_uint32(log_n)))


def batch_invert_device(ctx, col):
    """ff::BatchInvert on a device column, in place."""
    _check(lib().zkhip_batch_invert_device(ctx.h, C.c_void_p(col.data_ptr()), C.c_size_t(col.shape[0])))


def eval_polynomials_device(ctx, polys, x):
    """arithmetic::eval_polynomial for a batch of device polynomials of equal length -> (npolys, 4) device tensor."""
    out = ctx.empty(len(polys))
    if polys:
        _check(lib().zkhip_eval_polynomial_device(ctx.h, _ptr_array(polys), C.c_size_t(len(polys)), C.c_size_t(polys[0].shape[0]),
                                                  _p(_u64(x)), C.c_void_p(out.data_ptr())))
    return out


def eval_polynomials_at_device(ctx, polys, xs):
    """one point per polynomial: xs is an (npolys, 4) host array"""
    out = ctx.empty(len(polys))
    if polys:
        xs = _u64(xs).reshape(len(polys), 4)
        _check(lib().zkhip_eval_polynomials_at_device(ctx.h, _ptr_array(polys), C.c_size_t(len(polys)), C.c_size_t(polys[0].shape[0]), _p(xs),
                                                      C.c_void_p(out.data_ptr())))
    return out


def grand_products_device(ctx, k, beta, gamma, blinding_factors, values, sigmas, chunk_len, perm_blinding, lookups, lookup_blinding):
    """permutation::commit + every lookup's commit_product behind one batch inversion.
    lookups: list of (compressed_input, compressed_table, permuted_input, permuted_table).  -> (perm z list, lookup z list)"""
    nsets = -(-len(values) // chunk_len) if values else 0
    pz = [ctx.empty(1 << k) for _ in range(nsets)]
    lz = [ctx.empty(1 << k) for _ in lookups]
    none = C.c_void_p(0)
    cols = list(zip(*lookups)) if lookups else [[], [], [], []]
    _check(lib().zkhip_grand_products_device(
        ctx.h, C.c_uint32(k), _p(_u64(beta)), _p(_u64(gamma)), C.c_uint32(blinding_factors),
        _ptr_array(values), _ptr_array(sigmas), C.c_size_t(len(values)), C.c_uint32(max(1, chunk_len)),
        C.c_void_p(perm_blinding.data_ptr()) if nsets else none, _ptr_array(pz),
        C.c_size_t(len(lookups)), _ptr_array(list(cols[0])), _ptr_array(list(cols[1])), _ptr_array(list(cols[2])), _ptr_array(list(cols[3])),
        C.c_void_p(lookup_blinding.data_ptr()) if lookups else none, _ptr_array(lz)))
    return pz, lz


def permutation_products_device(ctx, k, values, sigmas, chunk_len, beta, gamma, blinding_factors, blinding):
    """permutation::prover::commit: list of z columns (Lagrange form).  blinding: (nsets * bf, 4) device tensor."""
    nsets = -(-len(values) // chunk_len)
    zs = [ctx.empty(1 << k) for _ in range(nsets)]
    if values:
        _check(lib().zkhip_permutation_products_device(ctx.h, C.c_uint32(k), _ptr_array(values), _ptr_array(sigmas), C.c_size_t(len(values)),
                                                       C.c_uint32(chunk_len), _p(_u64(beta)), _p(_u64(gamma)), C.c_uint32(blinding_factors),
                                                       C.c_void_p(blinding.data_ptr()), _ptr_array(zs)))
    return zs


def permute_expression_pair_device(ctx, k, blinding_factors, cin, ctab, blind_in, blind_tab):
    """lookup::prover::permute_expression_pair on device columns; raises ConstraintSystemFailure like halo2 does."""
    pin, ptab = ctx.empty(1 << k), ctx.empty(1 << k)
    rc = lib().zkhip_permute_expression_pair_device(ctx.h, C.c_uint32(k), C.c_uint32(blinding_factors), C.c_void_p(cin.data_ptr()),
                                                    C.c_void_p(ctab.data_ptr()), C.c_void_p(blind_in.data_ptr()),
                                                    C.c_void_p(blind_tab.data_ptr()), C.c_void_p(pin.data_ptr()), C.c_void_p(ptab.data_ptr()))
    if rc == ECONSTRAINT:
        raise ConstraintSystemFailure(lib().zkhip_last_error().decode())
    _check(rc)
    return pin, ptab


def linear_combination_device(ctx, polys, coeffs, low=None):
    """sum_j coeffs[j] * polys[j] - low; coeffs (npolys, 4) and low (nlow, 4) are host ABI arrays."""
    n = polys[0].shape[0]
    out = ctx.empty(n)
    coeffs = _u64(coeffs).reshape(len(polys), 4)
    nlow = 0 if low is None else len(low)
    low_a = _u64(low).reshape(nlow, 4) if nlow else None
    _check(lib().zkhip_linear_combination_device(ctx.h, C.c_size_t(n), _ptr_array(polys), C.c_size_t(len(polys)), _p(coeffs),
                                                 _p(low_a) if nlow else None, C.c_size_t(nlow), C.c_void_p(out.data_ptr())))
    return out


def divide_by_linear_device(ctx, srcs, roots):
    """-> new polynomials srcs[j] / (X - roots[j]); roots: host (len(srcs), 4) ABI array"""
    if not srcs:
        return []
    n = srcs[0].shape[0]
    outs = [ctx.empty(n) for _ in srcs]
    _check(lib().zkhip_divide_by_linear_device(ctx.h, C.c_size_t(n), _ptr_array(srcs), _ptr_array(outs), C.c_size_t(len(srcs)),
                                               _p(_u64(roots).reshape(len(srcs), 4))))
    return outs


def kate_division_device(ctx, polys, roots):
    """In place: polys[j] /= prod (X - r) for r in roots[j] (each a list of host ABI elements)."""
    if not polys:
        return
    n = polys[0].shape[0]
    counts = np.array([len(r) for r in roots], dtype=np.uint32)
    flat = _u64(np.concatenate([_u64(r).reshape(-1, 4) for r in roots if len(r)])) if counts.sum() else np.zeros((1, 4), dtype=np.uint64)
    _check(lib().zkhip_kate_division_device(ctx.h, C.c_size_t(n), _ptr_array(polys), C.c_size_t(len(polys)), _p(counts), _p(flat)))


HOST_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
WRITE_POINT_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64))
SQUEEZE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint64))
WRITE_SCALAR_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint64))


class ZkTranscript(C.Structure):
    _fields_ = [("user", C.c_void_p), ("write_point", WRITE_POINT_FN), ("squeeze_challenge", SQUEEZE_FN), ("write_scalar", WRITE_SCALAR_FN),
                ("common_scalar", WRITE_SCALAR_FN)]


def make_transcript(write_point, squeeze_challenge, write_scalar=None, common_scalar=None):
    """zk_transcript over Python callables: write_point(bytes32, xy (8,) uint64), squeeze_challenge() -> 4 ABI limbs,
    write_scalar / common_scalar((4,) uint64 ABI limbs).  Keep the returned object alive for the duration of the call."""
    def _wp(user, b, xy):
        write_point(C.string_at(b, 32), np.frombuffer(C.string_at(xy, 64), dtype=np.uint64))

    def _sq(user, out):
        C.memmove(out, np.ascontiguousarray(squeeze_challenge(), dtype=np.uint64).ctypes.data, 32)

    def _ws(user, sc):
        write_scalar(np.frombuffer(C.string_at(sc, 32), dtype=np.uint64))

    def _cs(user, sc):
        common_scalar(np.frombuffer(C.string_at(sc, 32), dtype=np.uint64))

    return ZkTranscript(None, WRITE_POINT_FN(_wp), SQUEEZE_FN(_sq), WRITE_SCALAR_FN(_ws) if write_scalar else WRITE_SCALAR_FN(),
                        WRITE_SCALAR_FN(_cs) if common_scalar else WRITE_SCALAR_FN())


class LibTranscript:
    """One of the library's ready-made transcripts (host code; usable without a GPU):
      "blake2b"  halo2's Blake2bWrite                      zkhip_blake2b_transcript_*
      "evm"      snark-verifier's EvmTranscript (Keccak)   zkhip_evm_transcript_*     gen_evm_proof_shplonk, /root/reference/src/bin/cli.rs:519
      "poseidon" snark-verifier's PoseidonTranscript       zkhip_poseidon_transcript_* gen_snark_shplonk, src/helpers.rs:233,299
    `callbacks` is the zk_transcript* zkhip_create_proof takes; the methods drive the same object from Python (the Python schedule)."""

    POINT_BYTES = {"blake2b": 32, "evm": 64, "poseidon": 32}

    def __init__(self, kind="blake2b"):
        if kind not in self.POINT_BYTES:
            raise ValueError(kind)
        self.kind = kind
        self._pfx = f"zkhip_{kind}_transcript_"
        self.h = C.c_void_p(self._fn("new")())
        self.callbacks = C.c_void_p(self._fn("callbacks")(self.h))
        self._cb = ZkTranscript.from_address(self.callbacks.value)

    def _fn(self, name):
        f = getattr(lib(), self._pfx + name)
        if name in ("new", "callbacks"):
            f.restype = C.c_void_p
        elif name in ("proof", "points", "challenges"):
            f.restype = C.c_size_t
        return f

    def __del__(self):
        if getattr(self, "h", None) and self.h.value and lib is not None:
            self._fn("free")(self.h)
            self.h = C.c_void_p()

    def proof(self):
        p = C.POINTER(C.c_uint8)()
        n = self._fn("proof")(self.h, C.byref(p))
        return C.string_at(p, n) if n else b""

    def points(self):
        p = C.POINTER(C.c_uint64)()
        n = self._fn("points")(self.h, C.byref(p))
        return np.frombuffer(C.string_at(p, n * 64), dtype=np.uint64).reshape(n, 8) if n else np.zeros((0, 8), dtype=np.uint64)

    def challenges(self):
        p = C.POINTER(C.c_uint64)()
        n = self._fn("challenges")(self.h, C.byref(p))
        return np.frombuffer(C.string_at(p, n * 32), dtype=np.uint64).reshape(n, 4) if n else np.zeros((0, 4), dtype=np.uint64)

    # ---- the same transcript driven from Python
    def write_point(self, xy, bytes32=None):
        xy = _u64(xy)
        b = bytes32 if bytes32 is not None else g1_to_bytes(xy)
        self._cb.write_point(self._cb.user, (C.c_uint8 * 32)(*b), xy.ctypes.data_as(C.POINTER(C.c_uint64)))

    def write_scalar(self, limbs):
        limbs = _u64(limbs)
        self._cb.write_scalar(self._cb.user, limbs.ctypes.data_as(C.POINTER(C.c_uint64)))

    def common_scalar(self, limbs):
        limbs = _u64(limbs)
        self._cb.common_scalar(self._cb.user, limbs.ctypes.data_as(C.POINTER(C.c_uint64)))

    def squeeze_limbs(self):
        out = (C.c_uint64 * 4)()
        self._cb.squeeze_challenge(self._cb.user, out)
        return np.array(list(out), dtype=np.uint64)


class NativeTranscript(LibTranscript):
    """The library's Blake2bWrite (zkhip_blake2b_transcript_*): no Python in the proof's critical path."""

    def __init__(self):
        super().__init__("blake2b")


class EvmTranscript(LibTranscript):
    """The library's EvmTranscript (Keccak-256; snark-verifier's transcript for EVM proofs)."""

    def __init__(self):
        super().__init__("evm")


class PoseidonTranscript(LibTranscript):
    """The library's PoseidonTranscript (snark-verifier's native transcript: every gen_snark_shplonk proof of the reference)."""

    def __init__(self):
        super().__init__("poseidon")


def poseidon_permute(state_limbs, plain=False):
    """the bare Poseidon permutation on 3 ABI elements ((3, 4) uint64) -> (3, 4); plain: textbook rounds instead of the sparse form"""
    st = _u64(state_limbs).reshape(3, 4).copy()
    (lib().zkhip_poseidon_permute_plain if plain else lib().zkhip_poseidon_permute)(_p(st))
    return st


def keccak256(data, pad=0x01):
    out = (C.c_uint8 * 32)()
    lib().zkhip_keccak256(data, C.c_size_t(len(data)), C.c_uint8(pad), out)
    return bytes(out)


class ZkProvingKey(C.Structure):
    _fields_ = [("k", C.c_uint32), ("cs_degree", C.c_uint32), ("blinding_factors", C.c_uint32),
                ("n_fixed", C.c_uint32), ("n_advice", C.c_uint32), ("n_instance", C.c_uint32), ("n_lookups", C.c_uint32),
                ("n_perm_columns", C.c_uint32),
                ("g", C.c_void_p), ("g_lagrange", C.c_void_p), ("domain", C.c_void_p),
                ("fixed_lagrange", C.c_void_p), ("fixed_coeff", C.c_void_p), ("fixed_cosets", C.c_void_p),
                ("sigma_lagrange", C.c_void_p), ("sigma_coeff", C.c_void_p), ("sigma_cosets", C.c_void_p),
                ("l0", C.c_void_p), ("l_last", C.c_void_p), ("l_active_row", C.c_void_p),
                ("custom_gates", ZkGraph), ("lookup_graphs", C.c_void_p), ("lookup_input_compress", C.c_void_p),
                ("lookup_table_compress", C.c_void_p), ("lookup_input_advice_column", C.c_void_p),
                ("lookup_table_fixed_column", C.c_void_p), ("key_id", C.c_uint64), ("perm_column_type", C.c_void_p),
                ("perm_column_index", C.c_void_p),
                ("n_advice_queries", C.c_uint32), ("n_fixed_queries", C.c_uint32),
                ("advice_query_column", C.c_void_p), ("advice_query_rotation", C.c_void_p),
                ("fixed_query_column", C.c_void_p), ("fixed_query_rotation", C.c_void_p),
                ("delta", C.c_uint64 * 4), ("vk_transcript_repr", C.c_void_p),
                ("advice_column_phase", C.c_void_p), ("n_challenges", C.c_uint32), ("challenge_phase", C.c_void_p)]


class ZkProofOut(C.Structure):
    _fields_ = [("d_h", C.c_void_p), ("evals", C.c_void_p), ("eval_poly", C.c_void_p), ("eval_rotation", C.c_void_p),
                ("eval_write_order", C.c_void_p), ("evals_cap", C.c_size_t), ("n_evals", C.c_size_t)]


class ZkBlinding(C.Structure):
    _fields_ = [("lookup_permuted", C.c_void_p), ("perm_z", C.c_void_p), ("lookup_z", C.c_void_p), ("random_poly", C.c_void_p),
                ("on_host", C.c_int)]


ADVICE_PHASE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_void_p))


class ZkProofInputs(C.Structure):
    _fields_ = [("advice", C.c_void_p), ("advice_on_host", C.c_int), ("d_instance", C.c_void_p), ("instance_values", C.c_void_p),
                ("instance_len", C.c_void_p), ("blinding", C.c_void_p), ("blinding_seed", C.c_uint64),
                ("advice_phase", ADVICE_PHASE_FN), ("advice_phase_user", C.c_void_p)]


def shplonk_open(ctx, params, polys, query_poly, query_points, query_evals, write_point, squeeze_challenge):
    """ProverSHPLONK::create_proof in the library.  polys: device polynomials; query_poly: index per query; query_points /
    query_evals: (nq, 4) ABI arrays; write_point(bytes32, xy (8,) uint64 array); squeeze_challenge() -> (4,) ABI limbs.
    Returns (h1_xy, h2_xy)."""
    nq = len(query_poly)
    qp = np.ascontiguousarray(query_poly, dtype=np.uint32)
    pts = _u64(query_points).reshape(nq, 4)
    evs = _u64(query_evals).reshape(nq, 4)

    t = make_transcript(write_point, squeeze_challenge)
    h1, h2 = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    _check(lib().zkhip_shplonk_open(ctx.h, params.g, C.c_size_t(polys[0].shape[0]), _ptr_array(polys), C.c_size_t(len(polys)), _p(qp), _p(pts),
                                    _p(evs), C.c_size_t(nq), C.byref(t), _p(h1), _p(h2)))
    return h1, h2


def lookup_product_device(ctx, k, cin, ctab, pin, ptab, beta, gamma, blinding_factors, blinding):
    z = ctx.empty(1 << k)
    _check(lib().zkhip_lookup_product_device(ctx.h, C.c_uint32(k), C.c_void_p(cin.data_ptr()), C.c_void_p(ctab.data_ptr()),
                                             C.c_void_p(pin.data_ptr()), C.c_void_p(ptab.data_ptr()), _p(_u64(beta)), _p(_u64(gamma)),
                                             C.c_uint32(blinding_factors), C.c_void_p(blinding.data_ptr()), C.c_void_p(z.data_ptr())))
    return z


def _ptr_array(tensors):
    return (C.c_void_p * max(1, len(tensors)))(*[t.data_ptr() for t in tensors])


_FQ = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
_FR = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
# BN254 G2 generator (EIP-197 / halo2curves bn256 G2 [UPSTREAM-RECALL]); Fq2 = Fq[u] / (u^2 + 1), curve y^2 = x^3 + 3 / (9 + u)
_G2X = (10857046999023057135944570762232829481370756359578518086990519993285655852781,
        11559732032986387107991004021392285783925812861821192530917403151452391805634)
_G2Y = (8495653923123431417604973247489272438418190587263600148770280649306958101930,
        4082367875863433681332203403145435568316851327593401208105741076214120093531)


def _g2_setup_bytes(s_int):
    """ParamsKZG::setup's G2 half on the host (big-int Fq2 arithmetic; setup time only): g2 and [s] g2 as 2 x 128 RawBytes
    (x.c0, x.c1, y.c0, y.c1: 4 LE u64 Montgomery limbs each).  Only the verifier's pairing uses them; the file format carries them."""
    q = _FQ
    mul = lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)
    add = lambda a, b: ((a[0] + b[0]) % q, (a[1] + b[1]) % q)
    sub = lambda a, b: ((a[0] - b[0]) % q, (a[1] - b[1]) % q)

    def inv(a):
        d = pow((a[0] * a[0] + a[1] * a[1]) % q, q - 2, q)
        return (a[0] * d % q, (-a[1]) * d % q)

    b2 = mul((3, 0), inv((9, 1)))
    assert mul(_G2Y, _G2Y) == add(mul(mul(_G2X, _G2X), _G2X), b2), "G2 generator not on the twist"

    def padd(p1, p2):      # affine, None = identity
        if p1 is None:
            return p2
        if p2 is None:
            return p1
        if p1[0] == p2[0]:
            if add(p1[1], p2[1]) == (0, 0):
                return None
            lam = mul(mul((3, 0), mul(p1[0], p1[0])), inv(add(p1[1], p1[1])))
        else:
            lam = mul(sub(p2[1], p1[1]), inv(sub(p2[0], p1[0])))
        x3 = sub(sub(mul(lam, lam), p1[0]), p2[0])
        return (x3, sub(mul(lam, sub(p1[0], x3)), p1[1]))

    acc, base, e = None, (_G2X, _G2Y), s_int % _FR
    while e:
        if e & 1:
            acc = padd(acc, base)
        base = padd(base, base)
        e >>= 1
    out = b""
    for pt in ((_G2X, _G2Y), acc):
        for c in (pt[0][0], pt[0][1], pt[1][0], pt[1][1]) if pt is not None else (0, 0, 0, 0):
            out += (c * (1 << 256) % q).to_bytes(32, "little")
    return out


class ParamsKZG:
    """ParamsKZG<Bn256> restricted to what the prover's commitments use: g and g_lagrange."""

    def __init__(self, ctx, k, g=None, g_lagrange=None):
        self.ctx, self.k, self.n = ctx, k, 1 << k
        self.g, self.g_lagrange = g, g_lagrange

    @classmethod
    def setup(cls, ctx, k, s):
        """ParamsKZG::setup(k, rng) with the trapdoor given (Montgomery Fr limbs)."""
        g, gl = C.c_void_p(), C.c_void_p()
        if ctx.world > 1 and ctx.shard_points:      # a communicator on the context: this rank builds the window tables of its point range only
            first, count = ctx.shard_range(1 << k)
            _check(lib().zkhip_kzg_setup_range(ctx.h, C.c_uint32(k), _p(_u64(s)), C.c_size_t(first), C.c_size_t(count), C.byref(g), C.byref(gl)))
        else:
            _check(lib().zkhip_kzg_setup(ctx.h, C.c_uint32(k), _p(_u64(s)), C.byref(g), C.byref(gl)))
        p = cls(ctx, k, g, gl)
        sl = _u64(s)
        s_int = sum(int(sl[i]) << (64 * i) for i in range(4)) * pow(1 << 256, -1, _FR) % _FR
        p.g2_bytes = _g2_setup_bytes(s_int)      # the real g2 and [s] g2: a file written from these params is a complete ParamsKZG
        return p

    @classmethod
    def from_bases(cls, ctx, k, g_xy=None, g_lagrange_xy=None):
        out = []
        for b in (g_xy, g_lagrange_xy):
            if b is None:
                out.append(None)
                continue
            b = _u64(b).reshape(-1, 8)
            h = C.c_void_p()
            _check(lib().zkhip_srs_load(ctx.h, _p(b), C.c_size_t(b.shape[0]), C.byref(h)))
            out.append(h)
        return cls(ctx, k, out[0], out[1])

    # halo2_proofs ParamsKZG::write / ::read (poly/kzg/commitment.rs, SerdeFormat::RawBytes) [UPSTREAM-RECALL]: the file the
    # reference keeps under PARAMS_DIR as kzg_bn254_{k}.srs (/root/reference/src/bin/cli.rs:222): k as u32 LE, g (n points),
    # g_lagrange (n points), g2 and s_g2; a G1 point is its x and y as 4 LE u64 Montgomery limbs each — byte for byte the
    # form zkhip_srs_load takes — and a G2 point 128 bytes, which the prover never touches (kept opaque for write()).
    @classmethod
    def read(cls, ctx, path):
        with open(path, "rb") as f:
            k = int.from_bytes(f.read(4), "little")
            if not 1 <= k <= 28:
                raise ZkhipError(f"{path}: k = {k} is not a KZG parameter file")
            n = 1 << k
            if ctx.world > 1 and ctx.shard_points:      # this rank's slice of both bases only
                first, count = ctx.shard_range(n)
                hs = []
                for b in range(2):
                    f.seek(4 + (b * n + first) * 64)
                    raw = f.read(count * 64)
                    if len(raw) != count * 64:
                        raise ZkhipError(f"{path}: truncated")
                    h = C.c_void_p()
                    _check(lib().zkhip_srs_load_range(ctx.h, _p(np.frombuffer(raw, dtype="<u8").copy()), C.c_size_t(n), C.c_size_t(first),
                                                      C.c_size_t(count), C.byref(h)))
                    hs.append(h)
                f.seek(4 + 2 * n * 64)
                g2 = f.read(256)
                p = cls(ctx, k, hs[0], hs[1])
                p.g2_bytes = g2
                return p
            raw = f.read(2 * n * 64)
            g2 = f.read(256)
        if len(raw) != 2 * n * 64 or len(g2) != 256:
            raise ZkhipError(f"{path}: truncated ({len(raw)} point bytes, {len(g2)} G2 bytes)")
        pts = np.frombuffer(raw, dtype="<u8").reshape(2, n, 8)
        p = cls.from_bases(ctx, k, g_xy=pts[0], g_lagrange_xy=pts[1])
        p.g2_bytes = g2
        return p

    def write(self, path):
        if self.range()[1] != self.range()[2]:
            raise ZkhipError("ParamsKZG.write: these params are one rank's shard of the SRS")
        if getattr(self, "g2_bytes", None) is None or len(self.g2_bytes) != 256:
            raise ZkhipError("ParamsKZG.write: these params carry no G2 points (loaded from bare bases); refusing to write an SRS file "
                             "with an identity g2 / s_g2")
        with open(path, "wb") as f:
            f.write(int(self.k).to_bytes(4, "little"))
            for h in (self.g, self.g_lagrange):
                for first in range(0, self.n, 1 << 16):
                    cnt = min(1 << 16, self.n - first)
                    f.write(self.read_bases(h, first, cnt).astype("<u8").tobytes())
            f.write(self.g2_bytes)

    def range(self):
        """(first, count, n_total): the global point range this handle's tables hold (the whole SRS unless sharded)"""
        a, b, c_ = C.c_size_t(), C.c_size_t(), C.c_size_t()
        lib().zkhip_srs_range(self.g if self.g is not None else self.g_lagrange, C.byref(a), C.byref(b), C.byref(c_))
        return a.value, b.value, c_.value

    def window(self):
        c, w = C.c_uint32(), C.c_uint32()
        lib().zkhip_srs_window(self.g if self.g is not None else self.g_lagrange, C.byref(c), C.byref(w))
        return c.value, w.value

    def read_bases(self, which, first, count):
        out = np.zeros((count, 8), dtype=np.uint64)
        _check(lib().zkhip_srs_read(self.ctx.h, which, C.c_size_t(first), C.c_size_t(count), _p(out)))
        return out

    def _msm_host(self, srs, scalars):
        scalars = _u64(scalars).reshape(-1, 4)
        out = np.zeros(12, dtype=np.uint64)
        _check(lib().zkhip_msm_g1(self.ctx.h, srs, _p(scalars), C.c_size_t(scalars.shape[0]), _p(out)))
        return out

    def commit_batch_host(self, polys, lagrange=False):
        """several HOST columns of one length in one call (zkhip_msm_g1_batch) -> (ncols, 12) uint64 array of normalised Jacobian sums"""
        cols = [_u64(p_).reshape(-1, 4) for p_ in polys]
        n = cols[0].shape[0] if cols else 0
        assert all(c.shape[0] == n for c in cols)
        out = np.zeros((len(cols), 12), dtype=np.uint64)
        ptrs = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
        _check(lib().zkhip_msm_g1_batch(self.ctx.h, self.g_lagrange if lagrange else self.g, ptrs, C.c_size_t(len(cols)), C.c_size_t(n), _p(out)))
        return out

    def commit(self, poly):
        """ParamsKZG::commit(poly): MSM over g[..len] -> G1 (Jacobian, 12 u64)."""
        return self._msm_host(self.g, poly)

    def commit_lagrange(self, poly):
        return self._msm_host(self.g_lagrange, poly)

    def commit_batch_device(self, cols, lagrange=False, n=None, first=0):
        """ncols device columns -> (ncols, 12) device tensor of Jacobian sums over points [first, first+n).
        `lagrange` is one bool for the batch or one per column (ParamsKZG::commit_lagrange vs commit)."""
        flags = list(lagrange) if isinstance(lagrange, (list, tuple)) else [bool(lagrange)] * len(cols)
        srs = (C.c_void_p * len(cols))(*[(self.g_lagrange if f else self.g).value for f in flags])
        n = n if n is not None else cols[0].shape[0] - first
        out = self.ctx.empty(len(cols), 12)
        _check(lib().zkhip_msm_g1_multi_device(self.ctx.h, srs, _ptr_array(cols), C.c_size_t(len(cols)), C.c_size_t(first),
                                               C.c_size_t(n), C.c_void_p(out.data_ptr())))
        return out

    def free(self):
        for h in (self.g, self.g_lagrange):
            if h is not None:
                lib().zkhip_srs_free(self.ctx.h, h)
        self.g = self.g_lagrange = None


def best_multiexp(ctx, coeffs, bases_xy):
    """halo2curves msm::best_multiexp(coeffs, bases) for ad-hoc bases (loads them, then one MSM)."""
    p = ParamsKZG.from_bases(ctx, 0, g_xy=bases_xy)
    try:
        return p.commit(coeffs)
    finally:
        p.free()


def g1_to_affine(xyz):
    out = np.zeros(8, dtype=np.uint64)
    lib().zkhip_g1_to_affine(_p(_u64(xyz)), _p(out))
    return out


def g1_batch_to_affine(xyz_rows):
    """(n, 12) Jacobian rows -> (n, 8) affine rows with one field inversion (Curve::batch_normalize)."""
    xyz_rows = _u64(xyz_rows).reshape(-1, 12)
    out = np.zeros((xyz_rows.shape[0], 8), dtype=np.uint64)
    lib().zkhip_g1_batch_to_affine(_p(xyz_rows), C.c_size_t(xyz_rows.shape[0]), _p(out))
    return out


def commitments_read(ctx, d_xyz):
    """(ncols, 12) device Jacobian sums -> [(affine (8,) u64, 32 compressed bytes)] with one sync and one inversion"""
    n = d_xyz.shape[0]
    out = np.zeros((n, 8), dtype=np.uint64)
    byts = (C.c_uint8 * (32 * n))()
    _check(lib().zkhip_commitments_read(ctx.h, C.c_void_p(d_xyz.data_ptr()), C.c_size_t(n), _p(out), byts))
    raw = bytes(byts)
    return [(out[j], raw[32 * j:32 * j + 32]) for j in range(n)]


def g1_add(a, b):
    out = np.zeros(12, dtype=np.uint64)
    lib().zkhip_g1_add(_p(_u64(a)), _p(_u64(b)), _p(out))
    return out


def g1_to_bytes(xy):
    out = (C.c_uint8 * 32)()
    lib().zkhip_g1_to_bytes(_p(_u64(xy)), out)
    return bytes(out)


class EvaluationDomain:
    """halo2_proofs::poly::EvaluationDomain::new(j, k)."""

    def __init__(self, ctx, j, k, g_coset=None):
        self.ctx = ctx
        self.h = C.c_void_p()
        _check(lib().zkhip_domain_new(ctx.h, C.c_uint32(j), C.c_uint32(k), _p(_u64(g_coset)) if g_coset is not None else None,
                                      C.byref(self.h)))
        self.k = k
        self.n = 1 << k
        self.extended_k = lib().zkhip_domain_extended_k(self.h)
        self.extended_n = 1 << self.extended_k
        self.quotient_poly_degree = lib().zkhip_domain_quotient_poly_degree(self.h)
        self.omega, self.extended_omega, self.g_coset = (np.zeros(4, dtype=np.uint64) for _ in range(3))
        lib().zkhip_domain_constants(self.h, _p(self.omega), _p(self.extended_omega), _p(self.g_coset))

    def free(self):
        if self.h is not None and self.h.value:
            lib().zkhip_domain_free(self.ctx.h, self.h)
            self.h = C.c_void_p()

    # host-array forms (as the reference's Vec<F> callers see them)
    def lagrange_to_coeff(self, a):
        a = _u64(a).copy().reshape(self.n, 4)
        _check(lib().zkhip_lagrange_to_coeff(self.ctx.h, self.h, _p(a)))
        return a

    def coeff_to_extended(self, coeffs):
        coeffs = _u64(coeffs).reshape(-1, 4)
        out = np.zeros((self.extended_n, 4), dtype=np.uint64)
        _check(lib().zkhip_coeff_to_extended(self.ctx.h, self.h, _p(coeffs), C.c_size_t(coeffs.shape[0]), _p(out)))
        return out

    def extended_to_coeff(self, a, full=False):
        a = _u64(a).copy().reshape(self.extended_n, 4)
        _check(lib().zkhip_extended_to_coeff(self.ctx.h, self.h, _p(a)))
        return a if full else a[: self.n * self.quotient_poly_degree]

    # device forms (asynchronous)
    def lagrange_to_coeff_device(self, polys):
        if not polys:
            return
        _check(lib().zkhip_lagrange_to_coeff_device(self.ctx.h, self.h, _ptr_array(polys), C.c_size_t(len(polys))))

    def coeff_to_lagrange_device(self, polys):
        _check(lib().zkhip_coeff_to_lagrange_device(self.ctx.h, self.h, _ptr_array(polys), C.c_size_t(len(polys))))

    def coeff_to_extended_device(self, polys, n_in=None):
        if not polys:
            return []
        outs = [self.ctx.empty(self.extended_n) for _ in polys]
        n_in = n_in if n_in is not None else polys[0].shape[0]
        _check(lib().zkhip_coeff_to_extended_device(self.ctx.h, self.h, _ptr_array(polys), C.c_size_t(n_in), _ptr_array(outs),
                                                    C.c_size_t(len(polys))))
        return outs

    def extended_to_coeff_device(self, polys):
        _check(lib().zkhip_extended_to_coeff_device(self.ctx.h, self.h, _ptr_array(polys), C.c_size_t(len(polys))))

    def divide_by_vanishing_poly_device(self, a):
        _check(lib().zkhip_divide_by_vanishing_device(self.ctx.h, self.h, C.c_void_p(a.data_ptr())))

    # the coset layout of the quotient (cosets.hip): q = quotient_poly_degree blocks of n, block r = extended rows r + E i
    def cosets(self):
        """-> (q, shifts): the coset generators s_r = g w_ext^r as a host (q, 4) ABI array; raises when q >= 2^(extended_k - k)"""
        q = C.c_uint32()
        _check(lib().zkhip_domain_cosets(self.ctx.h, self.h, C.byref(q), None))
        shifts = np.zeros((q.value, 4), dtype=np.uint64)
        _check(lib().zkhip_domain_cosets(self.ctx.h, self.h, C.byref(q), _p(shifts)))
        return q.value, shifts

    def coeff_to_cosets_device(self, polys):
        if not polys:
            return []
        outs = [self.ctx.empty(self.n * self.quotient_poly_degree) for _ in polys]
        _check(lib().zkhip_coeff_to_cosets_device(self.ctx.h, self.h, _ptr_array(polys), _ptr_array(outs), C.c_size_t(len(polys))))
        return outs

    def cosets_to_pieces_device(self, vals):
        """vals: q n numerator values in the coset layout (overwritten) -> q n coefficients of numerator / (X^n - 1)"""
        out = self.ctx.empty(self.n * self.quotient_poly_degree)
        _check(lib().zkhip_cosets_to_pieces_device(self.ctx.h, self.h, C.c_void_p(vals.data_ptr()), C.c_void_p(out.data_ptr())))
        return out

    def evaluate_h_cosets(self, pack, first_row=0, n_rows=None):
        """the sweep over coset-layout columns (zkhip_evaluate_h_cosets_device) -> (n_rows, 4) device tensor"""
        n_rows = self.n * self.quotient_poly_degree - first_row if n_rows is None else n_rows
        out = self.ctx.empty(n_rows)
        _check(lib().zkhip_evaluate_h_cosets_device(self.ctx.h, self.h, C.byref(pack.args), C.c_size_t(first_row), C.c_size_t(n_rows),
                                                    C.c_void_p(out.data_ptr())))
        return out


class EvalhPack:
    """Marshals evaluate_h inputs (device column tensors + host graphs) into zk_evalh_args."""

    def __init__(self):
        self.keep = []

    def _ptrs(self, tensors):
        arr = _ptr_array(tensors)
        self.keep += [arr, tensors]
        return C.cast(arr, C.c_void_p)

    def graph(self, g, to_mont):
        # the marshalled form of a finished graph is cached on it (graphs are built once at keygen)
        key = (len(g.constants), len(g.rotations), len(g.calculations))
        cached = getattr(g, "_zk_packed", None)
        if cached is not None and cached[0] == key:
            consts, rots, code = cached[1]
        else:
            consts = _u64(to_mont(g.constants)).reshape(-1, 4)
            rots = np.array(g.rotations if g.rotations else [0], dtype=np.int32)
            code = np.array(g.code_words(), dtype=np.int32)
            g._zk_packed = (key, (consts, rots, code))
        self.keep += [consts, rots, code]
        return ZkGraph(consts.ctypes.data, rots.ctypes.data, code.ctypes.data, len(g.constants), len(g.rotations), len(code),
                       len(g.calculations), g.num_intermediates)

    def build(self, *, k, extended_k, cs_degree, blinding_factors, extended_omega, g_coset, delta, beta, gamma, theta, y,
              fixed, advice, instance, challenges, l0, l_last, l_active, gates_graph, perm_columns, sigma, perm_z,
              lookup_graphs, lookup_z, lookup_a, lookup_s, to_mont):
        a = ZkEvalhArgs()
        a.k, a.extended_k, a.cs_degree, a.blinding_factors = k, extended_k, cs_degree, blinding_factors
        for name, v in (("extended_omega", extended_omega), ("g_coset", g_coset), ("delta", delta), ("beta", beta),
                        ("gamma", gamma), ("theta", theta), ("y", y)):
            setattr(a, name, (C.c_uint64 * 4)(*[int(t) for t in v]))
        a.n_fixed, a.n_advice, a.n_instance = len(fixed), len(advice), len(instance)
        ch = _u64(challenges).reshape(-1, 4) if len(challenges) else np.zeros((1, 4), dtype=np.uint64)
        self.keep.append(ch)
        a.n_challenges = len(challenges)
        a.challenges = ch.ctypes.data
        a.fixed_cosets, a.advice_cosets, a.instance_cosets = self._ptrs(fixed), self._ptrs(advice), self._ptrs(instance)
        a.l0 = l0.data_ptr() if l0 is not None else None
        a.l_last = l_last.data_ptr() if l_last is not None else None
        a.l_active_row = l_active.data_ptr() if l_active is not None else None
        self.keep += [l0, l_last, l_active]
        a.custom_gates = self.graph(gates_graph, to_mont)
        tmap = {"advice": 0, "fixed": 1, "instance": 2}
        ptype = np.array([tmap[t] for t, _ in perm_columns] or [0], dtype=np.uint32)
        pidx = np.array([i for _, i in perm_columns] or [0], dtype=np.uint32)
        self.keep += [ptype, pidx]
        a.n_perm_columns, a.n_perm_sets = len(perm_columns), len(perm_z)
        a.perm_column_type, a.perm_column_index = ptype.ctypes.data, pidx.ctypes.data
        a.perm_sigma_cosets, a.perm_product_cosets = self._ptrs(sigma), self._ptrs(perm_z)
        a.n_lookups = len(lookup_graphs)
        garr = (ZkGraph * max(1, len(lookup_graphs)))(*[self.graph(g, to_mont) for g in lookup_graphs])
        self.keep.append(garr)
        a.lookup_graphs = C.cast(garr, C.c_void_p)
        a.lookup_product_cosets, a.lookup_input_cosets, a.lookup_table_cosets = (self._ptrs(lookup_z), self._ptrs(lookup_a),
                                                                                   self._ptrs(lookup_s))
        self.args = a
        return a


def evaluate_h_rows(ctx, pack, first_row, n_rows):
    """rows [first_row, first_row + n_rows) of evaluate_h (zkhip_evaluate_h_rows_device) -> (n_rows, 4) device tensor"""
    out = ctx.empty(n_rows)
    _check(lib().zkhip_evaluate_h_rows_device(ctx.h, C.byref(pack.args), C.c_size_t(first_row), C.c_size_t(n_rows), C.c_void_p(out.data_ptr())))
    return out


def evaluate_h(ctx, pack, extended_n):
    """Evaluator::evaluate_h on cosets already on the device -> (extended_n, 4) device tensor."""
    out = ctx.empty(extended_n)
    _check(lib().zkhip_evaluate_h_device(ctx.h, C.byref(pack.args), C.c_void_p(out.data_ptr())))
    return out
