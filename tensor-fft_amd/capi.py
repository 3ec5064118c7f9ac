"""ctypes binding of include/tfft.h (libtfft.so). No fallback of any kind."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libtfft.so"

TFFT_OK = 0
ERR_NAMES = {
    1: "TFFT_ERR_NOT_POW2", 2: "TFFT_ERR_TOO_SMALL", 3: "TFFT_ERR_MODE", 4: "TFFT_ERR_GEOMETRY",
    5: "TFFT_ERR_ARG", 6: "TFFT_ERR_DEVICE", 7: "TFFT_ERR_HIP", 8: "TFFT_ERR_WORKSPACE", 9: "TFFT_ERR_COMM",
}

# every symbol include/tfft.h declares (tests check that the library exports them all)
SYMBOLS = [
    "tfft_ref_create_plan", "tfft_device_check", "tfft_max_no_optin_shared_mem", "tfft_plan_create",
    "tfft_plan_destroy", "tfft_plan_num_launches", "tfft_plan_workspace_bytes", "tfft_plan_set_workspace",
    "tfft_exec", "tfft_plan_kernel_name", "tfft_plan_algorithmic_bytes", "tfft_plan_mfma_flops",
    "tfft_last_error", "tfft_version", "tfft_permute_twiddle", "tfft_exec_inverse", "tfft_deinterleave", "tfft_interleave",
    "tfft_plan2d_create", "tfft_plan2d_destroy", "tfft_plan2d_num_launches", "tfft_plan2d_workspace_bytes",
    "tfft_plan2d_set_workspace", "tfft_plan2d_exec", "tfft_plan2d_exec_inverse", "tfft_plan_describe",
    "tfft_variant_check", "tfft_plan_transposed_n2", "tfft_plan_cache_policy", "tfft_plan_default_variant", "tfft_synth_uniform",
    "tfft_dist_geometry_query", "tfft_dist_unique_id", "tfft_dist_comm_create", "tfft_dist_comm_create_all",
    "tfft_dist_comm_destroy", "tfft_dist_group_start", "tfft_dist_group_end", "tfft_dist_plan_create",
    "tfft_dist_plan_destroy", "tfft_dist_plan_geometry", "tfft_dist_plan_buffers", "tfft_dist_plan_set_buffers",
    "tfft_dist_exec_pre", "tfft_dist_exec_exchange", "tfft_dist_exec_post", "tfft_dist_exec",
    "tfft_copy_h2d", "tfft_copy_d2h", "tfft_plan_prepare", "tfft_plan_opts_init", "tfft_plan_opts_known_size",
    "tfft_dist_rccl_version", "tfft_dist_comm_info", "tfft_kernel_list",
    "tfft_tuning_load", "tfft_tuning_add", "tfft_tuning_clear", "tfft_tuning_query", "tfft_abi_version",
]
ABI_VERSION = 2                                               # TFFT_ABI_VERSION this binding's struct mirrors were written against

LAUNCH_PERSISTENT = 65535                                     # tfft_plan_opts.launch_iters
SCALE_SEQUENTIAL, SCALE_NONE, SCALE_ONCE = 0, 1, 2           # tfft_plan_opts.scale
ORDER_NATURAL, ORDER_TRANSPOSED = 0, 1                        # tfft_plan_opts.output_order
_SCALES = {"sequential": 0, "none": 1, "once": 2}
_ORDERS = {"natural": 0, "transposed": 1}


class TfftError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {message}")
        self.code = code
        self.message = message


class RefPlanStruct(ctypes.Structure):
    """tfft_ref_plan == struct Plan<Integer> of the reference (src/base/Plan.h:18-39)."""
    _fields_ = [
        ("fft_length", ctypes.c_uint64),
        ("amount_of_r16_steps", ctypes.c_int),
        ("amount_of_r2_steps", ctypes.c_int),
        ("base_fft_mode", ctypes.c_int),
        ("results_in_results", ctypes.c_int),
        ("base_fft_warps_per_block", ctypes.c_int),
        ("base_fft_blocksize", ctypes.c_int),
        ("base_fft_gridsize", ctypes.c_int),
        ("base_fft_shared_mem_in_bytes", ctypes.c_int),
        ("r16_warps_per_block", ctypes.c_int),
        ("r16_blocksize", ctypes.c_int),
        ("r16_gridsize", ctypes.c_int),
        ("r16_shared_mem_in_bytes", ctypes.c_int),
        ("r2_blocksize", ctypes.c_int),
    ]


class PlanOpts(ctypes.Structure):
    """tfft_plan_opts (include/tfft.h): struct_size first, the library reads exactly that many bytes."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("reserved_", ctypes.c_uint32),
        ("in_batch_stride", ctypes.c_uint64),
        ("out_batch_stride", ctypes.c_uint64),
        ("inner", ctypes.c_uint64),
        ("preserve_input", ctypes.c_int),
        ("variant", ctypes.c_int),
        ("scale", ctypes.c_int),
        ("output_order", ctypes.c_int),
        ("fourstep_n", ctypes.c_uint64),
        ("fourstep_col0", ctypes.c_uint64),
        ("launch_iters", ctypes.c_uint32),
        ("input_order", ctypes.c_int),
    ]


def _debug_requested():
    return os.environ.get("TFFT_DEBUG_VARIANTS") == "1" and os.environ.get("TFFT_USE_DEBUG_LIB") == "1"


class DistGeometry(ctypes.Structure):
    """tfft_dist_geometry (include/tfft.h)."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("reserved_", ctypes.c_uint32),
        ("n", ctypes.c_uint64), ("n1", ctypes.c_uint64), ("n2", ctypes.c_uint64), ("cols", ctypes.c_uint64),
        ("rows", ctypes.c_uint64), ("chunk", ctypes.c_uint64), ("world", ctypes.c_int), ("rank", ctypes.c_int),
        ("fused", ctypes.c_int), ("reorder", ctypes.c_int), ("local_passes", ctypes.c_int), ("slabs", ctypes.c_int),
    ]


    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = ctypes.sizeof(DistGeometry)       # OUT struct: the library fills at most this many bytes


DIST_ID_BYTES = 128
DIST_SELF_VIA_COMM, DIST_CALLER_BUFFERS, DIST_SLABS_2, DIST_SLABS_4 = 1, 2, 4, 8          # tfft_dist_plan_create flags


def lib_path():
    """libtfft.so. Only the measurement drivers under tools/ (which set TFFT_DEBUG_VARIANTS=1 and TFFT_USE_DEBUG_LIB=1
    before the first load) get build/libtfft_debug.so, the -DTFFT_DEBUG_KERNELS build with the timing-only kernels; the package
    directory holds nothing but the shipped library."""
    if _debug_requested():
        return os.path.join(os.path.dirname(_HERE), "build", "libtfft_debug.so")
    return os.path.join(_HERE, _LIB_NAME)


_lib = None


def load_library():
    """Loads libtfft.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` from the repository root.")
    # torch first: libtfft.so must bind to the HIP runtime (libamdhip64) that torch has loaded, because
    # device pointers and streams cross between the two. Loaded the other way round, the process ends up
    # with two HIP runtimes and the second one sees no device.
    import torch  # noqa: F401

    L = ctypes.CDLL(path)
    L.tfft_abi_version.restype = ctypes.c_int
    if L.tfft_abi_version() != ABI_VERSION:
        raise ImportError(f"{path} speaks ABI {L.tfft_abi_version()}, this binding was written against ABI {ABI_VERSION}: rebuild the library")
    vp, u64, ci = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int
    L.tfft_ref_create_plan.restype = ci
    L.tfft_ref_create_plan.argtypes = [u64, ci, ci, ci, ci, ctypes.POINTER(RefPlanStruct)]
    L.tfft_device_check.restype = ci
    L.tfft_device_check.argtypes = [ci]
    L.tfft_max_no_optin_shared_mem.restype = ci
    L.tfft_max_no_optin_shared_mem.argtypes = [ci]
    L.tfft_plan_create.restype = ci
    L.tfft_plan_create.argtypes = [u64, u64, ci, ctypes.POINTER(PlanOpts), ctypes.POINTER(vp)]
    L.tfft_plan_destroy.restype = None
    L.tfft_plan_destroy.argtypes = [vp]
    L.tfft_plan_num_launches.restype = ci
    L.tfft_plan_num_launches.argtypes = [vp]
    L.tfft_plan_workspace_bytes.restype = ctypes.c_size_t
    L.tfft_plan_workspace_bytes.argtypes = [vp]
    L.tfft_plan_set_workspace.restype = ci
    L.tfft_plan_set_workspace.argtypes = [vp, vp, ctypes.c_size_t]
    L.tfft_exec.restype = ci
    L.tfft_exec.argtypes = [vp, vp, vp, vp, vp, vp]
    L.tfft_deinterleave.restype = ci
    L.tfft_deinterleave.argtypes = [vp, vp, vp, u64, vp]
    L.tfft_interleave.restype = ci
    L.tfft_interleave.argtypes = [vp, vp, vp, u64, vp]
    L.tfft_exec_inverse.restype = ci
    L.tfft_exec_inverse.argtypes = [vp, vp, vp, vp, vp, vp]
    L.tfft_permute_twiddle.restype = ci
    L.tfft_permute_twiddle.argtypes = [vp, vp, vp, vp, u64, u64, u64, u64, u64, vp]
    L.tfft_plan2d_create.restype = ci
    L.tfft_plan2d_create.argtypes = [u64, u64, u64, ci, ctypes.POINTER(vp)]
    L.tfft_plan2d_destroy.restype = None
    L.tfft_plan2d_destroy.argtypes = [vp]
    L.tfft_plan2d_num_launches.restype = ci
    L.tfft_plan2d_num_launches.argtypes = [vp]
    L.tfft_plan2d_workspace_bytes.restype = ctypes.c_size_t
    L.tfft_plan2d_workspace_bytes.argtypes = [vp]
    L.tfft_plan2d_set_workspace.restype = ci
    L.tfft_plan2d_set_workspace.argtypes = [vp, vp, ctypes.c_size_t]
    L.tfft_plan2d_exec.restype = ci
    L.tfft_plan2d_exec.argtypes = [vp, vp, vp, vp, vp, vp]
    L.tfft_plan2d_exec_inverse.restype = ci
    L.tfft_plan2d_exec_inverse.argtypes = [vp, vp, vp, vp, vp, vp]
    L.tfft_plan_describe.restype = ci
    L.tfft_plan_describe.argtypes = [u64, u64, ci, ctypes.c_char_p, ctypes.c_size_t]
    L.tfft_variant_check.restype = ci
    L.tfft_variant_check.argtypes = [u64, u64, ci]
    L.tfft_plan_transposed_n2.restype = u64
    L.tfft_plan_transposed_n2.argtypes = [u64]
    L.tfft_plan_cache_policy.restype = ci
    L.tfft_plan_cache_policy.argtypes = [u64, u64, u64]
    L.tfft_plan_default_variant.restype = ci
    L.tfft_plan_default_variant.argtypes = [u64, u64, u64]
    L.tfft_synth_uniform.restype = ci
    L.tfft_synth_uniform.argtypes = [vp, vp, u64, u64, u64, u64, u64, vp]
    L.tfft_plan_kernel_name.restype = ctypes.c_char_p
    L.tfft_plan_kernel_name.argtypes = [vp]
    L.tfft_plan_algorithmic_bytes.restype = ctypes.c_double
    L.tfft_plan_algorithmic_bytes.argtypes = [vp]
    L.tfft_plan_mfma_flops.restype = ctypes.c_double
    L.tfft_plan_mfma_flops.argtypes = [vp]
    gp = ctypes.POINTER(DistGeometry)
    L.tfft_dist_geometry_query.restype = ci
    L.tfft_dist_geometry_query.argtypes = [u64, ci, ci, gp]
    L.tfft_dist_unique_id.restype = ci
    L.tfft_dist_unique_id.argtypes = [vp]
    L.tfft_dist_comm_create.restype = ci
    L.tfft_dist_comm_create.argtypes = [ci, ci, vp, ci, ctypes.POINTER(vp)]
    L.tfft_dist_comm_create_all.restype = ci
    L.tfft_dist_comm_create_all.argtypes = [ci, ctypes.POINTER(ci), ctypes.POINTER(vp)]
    L.tfft_dist_comm_destroy.restype = ci
    L.tfft_dist_comm_destroy.argtypes = [vp]
    L.tfft_dist_group_start.restype = ci
    L.tfft_dist_group_start.argtypes = []
    L.tfft_dist_group_end.restype = ci
    L.tfft_dist_group_end.argtypes = []
    L.tfft_dist_plan_create.restype = ci
    L.tfft_dist_plan_create.argtypes = [u64, ci, ci, ci, vp, ci, ctypes.POINTER(vp)]
    L.tfft_dist_plan_destroy.restype = None
    L.tfft_dist_plan_destroy.argtypes = [vp]
    L.tfft_dist_plan_geometry.restype = ci
    L.tfft_dist_plan_geometry.argtypes = [vp, gp]
    L.tfft_dist_plan_buffers.restype = ci
    L.tfft_dist_plan_buffers.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.tfft_dist_plan_set_buffers.restype = ci
    L.tfft_dist_plan_set_buffers.argtypes = [vp, vp, vp, vp, vp]
    L.tfft_dist_exec_pre.restype = ci
    L.tfft_dist_exec_pre.argtypes = [vp, vp, vp, vp]
    L.tfft_dist_exec_exchange.restype = ci
    L.tfft_dist_exec_exchange.argtypes = [vp, vp]
    L.tfft_dist_exec_post.restype = ci
    L.tfft_dist_exec_post.argtypes = [vp, vp, vp, vp]
    L.tfft_dist_exec.restype = ci
    L.tfft_dist_exec.argtypes = [vp, vp, vp, vp, vp, vp]
    L.tfft_copy_h2d.restype = ci
    L.tfft_copy_h2d.argtypes = [vp, vp, ctypes.c_size_t]
    L.tfft_copy_d2h.restype = ci
    L.tfft_copy_d2h.argtypes = [vp, vp, ctypes.c_size_t]
    L.tfft_plan_prepare.restype = ci
    L.tfft_plan_prepare.argtypes = [vp]
    L.tfft_plan_opts_init.restype = ci
    L.tfft_plan_opts_init.argtypes = [vp, ctypes.c_size_t]
    L.tfft_plan_opts_known_size.restype = ci
    L.tfft_plan_opts_known_size.argtypes = [ctypes.c_size_t]
    L.tfft_dist_comm_info.restype = ci
    L.tfft_dist_comm_info.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    L.tfft_dist_rccl_version.restype = ci
    L.tfft_dist_rccl_version.argtypes = [ctypes.POINTER(ci)]
    L.tfft_last_error.restype = ctypes.c_char_p
    L.tfft_last_error.argtypes = []
    L.tfft_version.restype = ctypes.c_char_p
    L.tfft_version.argtypes = []
    _lib = L
    return L


def last_error():
    return load_library().tfft_last_error().decode()


def _check(rc):
    if rc != TFFT_OK:
        raise TfftError(rc, last_error())


def plan_describe(n, inner=1, variant=0):
    """tfft_plan_describe: the pass decomposition a plan would get, as text. Host only, no GPU needed."""
    buf = ctypes.create_string_buffer(256)
    _check(load_library().tfft_plan_describe(int(n), int(inner), int(variant), buf, len(buf)))
    return buf.value.decode()


def kernel_list():
    """tfft_kernel_list: the column-kernel instantiations of the library's dispatch table (demangled names). Host only."""
    buf = ctypes.create_string_buffer(1 << 16)
    L = load_library()
    L.tfft_kernel_list.restype = ctypes.c_int
    L.tfft_kernel_list.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    rc = L.tfft_kernel_list(buf, len(buf))
    if rc < 0:
        raise TfftError(rc, last_error())
    names = buf.value.decode().split("\n")[:-1]
    assert len(names) == rc
    return names


def tuning_load(path):
    """tfft_tuning_load: tuner file -> process-wide plan wisdom; returns the number of lines taken. Host only."""
    L = load_library()
    L.tfft_tuning_load.restype = ctypes.c_int
    L.tfft_tuning_load.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    taken = ctypes.c_int(0)
    _check(L.tfft_tuning_load(os.fsencode(path), ctypes.byref(taken)))
    return taken.value


def tuning_add(n, batch, variant, launch_iters=0):
    L = load_library()
    L.tfft_tuning_add.restype = ctypes.c_int
    L.tfft_tuning_add.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32]
    _check(L.tfft_tuning_add(int(n), int(batch), int(variant), int(launch_iters)))


def tuning_clear():
    L = load_library()
    L.tfft_tuning_clear.restype = None
    L.tfft_tuning_clear()


def tuning_query(n, batch):
    """(variant, launch_iters) of the loaded tuner line that applies to (n, batch), or None."""
    L = load_library()
    L.tfft_tuning_query.restype = ctypes.c_int
    L.tfft_tuning_query.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint32)]
    v, it = ctypes.c_int(0), ctypes.c_uint32(0)
    return (v.value, it.value) if L.tfft_tuning_query(int(n), int(batch), ctypes.byref(v), ctypes.byref(it)) else None


def variant_check(n, inner=1, variant=0):
    """tfft_variant_check: raises TfftError unless `variant` is a value tfft_plan_create accepts. Host only."""
    _check(load_library().tfft_variant_check(int(n), int(inner), int(variant)))


def transposed_n2(n):
    """N2 of the TRANSPOSED output order (out[k1 * N2 + k2] = X[k1 + (N / N2) * k2]); 0 = natural order only."""
    return int(load_library().tfft_plan_transposed_n2(int(n)))


def plan_cache_policy(n, inner=1, batch=1):
    """tfft_plan_cache_policy: True = the column passes of this natural-order plan use plain (cached) accesses. Host only."""
    return bool(load_library().tfft_plan_cache_policy(int(n), int(inner), int(batch)))


def plan_default_variant(n, inner=1, batch=1):
    """tfft_plan_default_variant: the planner bits a variant-0 natural-order plan of this shape gets (0 unless the work is too small
    to fill the chip with the large-batch split). Host only."""
    return int(load_library().tfft_plan_default_variant(int(n), int(inner), int(batch)))


def ref_create_plan(fft_length, mode=0, base_fft_warps_per_block=8, r16_warps_per_block=8, r2_blocksize=256):
    """tfft_ref_create_plan -> (rc, RefPlanStruct, message). Host only, no GPU needed."""
    L = load_library()
    s = RefPlanStruct()
    rc = L.tfft_ref_create_plan(int(fft_length), int(mode), int(base_fft_warps_per_block),
                                int(r16_warps_per_block), int(r2_blocksize), ctypes.byref(s))
    return rc, s, last_error()


def device_check(device_id=0):
    _check(load_library().tfft_device_check(int(device_id)))


class TfftPlan:
    """Owning wrapper of tfft_plan. exec() takes torch CUDA half tensors (planar)."""

    def __init__(self, n, batch=1, device=0, in_batch_stride=0, out_batch_stride=0, preserve_input=False,
                 variant=0, inner=1, scale="sequential", output_order="natural", fourstep_n=0, fourstep_col0=0,
                 launch_iters=0, input_order="natural"):
        L = load_library()
        self._lib = L
        self._h = ctypes.c_void_p()
        scale = _SCALES[scale] if isinstance(scale, str) else int(scale)
        output_order = _ORDERS[output_order] if isinstance(output_order, str) else int(output_order)
        input_order = _ORDERS[input_order] if isinstance(input_order, str) else int(input_order)
        opts = PlanOpts(ctypes.sizeof(PlanOpts), 0, int(in_batch_stride), int(out_batch_stride), int(inner), int(bool(preserve_input)),
                        int(variant), scale, output_order, int(fourstep_n), int(fourstep_col0), int(launch_iters), input_order)
        self.scale, self.output_order, self.input_order = scale, output_order, input_order
        _check(L.tfft_plan_create(int(n), int(batch), int(device), ctypes.byref(opts), ctypes.byref(self._h)))
        self.n, self.batch, self.device, self.inner = int(n), int(batch), int(device), int(inner)
        self.in_batch_stride = int(in_batch_stride) or 2 * self.n * self.inner
        self.out_batch_stride = int(out_batch_stride) or 2 * self.n * self.inner
        self._ws = None

    def close(self):
        h = getattr(self, "_h", None)
        if h:
            self._h = None
            self._lib.tfft_plan_destroy(h)

    __del__ = close

    @property
    def num_launches(self):
        return self._lib.tfft_plan_num_launches(self._h)

    @property
    def workspace_bytes(self):
        return self._lib.tfft_plan_workspace_bytes(self._h)

    @property
    def kernel_name(self):
        return self._lib.tfft_plan_kernel_name(self._h).decode()

    @property
    def algorithmic_bytes(self):
        return self._lib.tfft_plan_algorithmic_bytes(self._h)

    @property
    def mfma_flops(self):
        return self._lib.tfft_plan_mfma_flops(self._h)

    def set_workspace(self, tensor):
        """Hands a torch CUDA tensor in as scratch (kept alive by the plan)."""
        self._ws = tensor
        _check(self._lib.tfft_plan_set_workspace(self._h, tensor.data_ptr(), tensor.numel() * tensor.element_size()))

    def exec_ptr(self, in_re, in_im, out_re, out_im, stream=0):
        _check(self._lib.tfft_exec(self._h, in_re, in_im, out_re, out_im, stream))

    def exec_inverse(self, in_re, in_im, out_re, out_im, stream=None):
        """(1/N) sum x[j] exp(+2 pi i jk/N): the forward transform on exchanged planes (tfft_exec_inverse)."""
        self.exec(in_im, in_re, out_im, out_re, stream)

    def exec(self, in_re, in_im, out_re, out_im, stream=None):
        import torch

        for t in (in_re, in_im, out_re, out_im):
            if not (t.is_cuda and t.dtype == torch.float16 and t.is_contiguous()):
                raise TfftError(5, "planes must be contiguous CUDA float16 tensors")
            if t.device.index != self.device:
                raise TfftError(5, "tensor on another device than the plan")
        need_in = (self.batch - 1) * self.in_batch_stride + self.n * self.inner
        need_out = (self.batch - 1) * self.out_batch_stride + self.n * self.inner
        if in_re.numel() < need_in or in_im.numel() < need_in or out_re.numel() < need_out or out_im.numel() < need_out:
            raise TfftError(5, "a plane is shorter than (batch-1)*stride + N")
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            self.exec_ptr(in_re.data_ptr(), in_im.data_ptr(), out_re.data_ptr(), out_im.data_ptr(), stream)


def permute_twiddle(in_re, in_im, out_re, out_im, a, b, c, n_tw=0, e0=0, stream=None):
    """out[b][a][c] = in[a][b][c] * w_n_tw^((e0+b)(a*C+c)) on torch CUDA half tensors (tfft_permute_twiddle)."""
    import torch

    for t in (in_re, in_im, out_re, out_im):
        if not (t.is_cuda and t.dtype == torch.float16 and t.is_contiguous() and t.numel() >= a * b * c):
            raise TfftError(5, "planes must be contiguous CUDA float16 tensors of at least A*B*C elements")
    dev = in_re.device.index
    if stream is None:
        stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        _check(load_library().tfft_permute_twiddle(in_re.data_ptr(), in_im.data_ptr(), out_re.data_ptr(),
                                                   out_im.data_ptr(), int(a), int(b), int(c), int(n_tw), int(e0), stream))


class TfftPlan2D:
    """2D transform of `batch` images [rows][cols] (fully planar: all RE images, then all IM images), result
    DFT2(x) / (rows * cols): tfft_plan2d_* of the C ABI, i.e. BASELINE config 4 as it describes it: a row pass over
    contiguous lines (N = cols, batch = batch * rows) and a column pass along the strided axis (N = rows,
    inner = cols). The reference has no 2D transform; this is its 1D path used twice."""

    def __init__(self, rows, cols, batch=1, device=0):
        self._lib = load_library()
        self.rows, self.cols, self.batch, self.device = int(rows), int(cols), int(batch), int(device)
        h = ctypes.c_void_p()
        _check(self._lib.tfft_plan2d_create(self.rows, self.cols, self.batch, self.device, ctypes.byref(h)))
        self._h = h
        self._ws = None

    @property
    def num_launches(self):
        return int(self._lib.tfft_plan2d_num_launches(self._h))

    @property
    def workspace_bytes(self):
        return int(self._lib.tfft_plan2d_workspace_bytes(self._h))

    def set_workspace(self, tensor):
        """Hand the plan a device buffer of at least workspace_bytes (kept referenced by this object)."""
        _check(self._lib.tfft_plan2d_set_workspace(self._h, tensor.data_ptr(), tensor.numel() * tensor.element_size()))
        self._ws = tensor

    def exec(self, in_re, in_im, out_re, out_im, stream=None):
        import torch

        n = self.batch * self.rows * self.cols
        for t in (in_re, in_im, out_re, out_im):
            if not (t.is_cuda and t.dtype == torch.float16 and t.is_contiguous() and t.numel() >= n):
                raise TfftError(5, "planes must be contiguous CUDA float16 tensors of batch*rows*cols elements")
        if self._ws is None:        # torch-owned scratch, so that nothing is hipMalloc'ed behind the caching allocator
            self.set_workspace(torch.empty(max(1, self.workspace_bytes // 2), dtype=torch.float16, device=in_re.device))
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _check(self._lib.tfft_plan2d_exec(self._h, in_re.data_ptr(), in_im.data_ptr(), out_re.data_ptr(),
                                              out_im.data_ptr(), stream))

    def exec_inverse(self, in_re, in_im, out_re, out_im, stream=None):
        """(1 / (rows cols)) sum x exp(+2 pi i (...)): the forward plan with the planes exchanged on both sides."""
        self.exec(in_im, in_re, out_im, out_re, stream)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tfft_plan2d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001 - interpreter shutdown
            pass


def dist_geometry(n, world, rank=0):
    """tfft_dist_geometry_query: the four-step split of a transform distributed over `world` ranks. Host only."""
    g = DistGeometry()
    _check(load_library().tfft_dist_geometry_query(int(n), int(world), int(rank), ctypes.byref(g)))
    return g


def dist_rccl_version():
    """ncclGetVersion of the RCCL bound by the library (e.g. 22703), for run reports."""
    v = ctypes.c_int(0)
    _check(load_library().tfft_dist_rccl_version(ctypes.byref(v)))
    return int(v.value)


def dist_unique_id():
    """An ncclUniqueId (bytes) from tfft_dist_unique_id: make it on one rank, carry it to the others."""
    buf = ctypes.create_string_buffer(DIST_ID_BYTES)
    _check(load_library().tfft_dist_unique_id(buf))
    return buf.raw


class DistComm:
    """Owning wrapper of an RCCL communicator created through the C ABI (tfft_dist_comm_create = ncclCommInitRank)."""

    def __init__(self, world, rank, unique_id, device):
        self._lib = load_library()
        if len(unique_id) != DIST_ID_BYTES:
            raise TfftError(5, "the unique id has to be 128 bytes")
        self._h = ctypes.c_void_p()
        _check(self._lib.tfft_dist_comm_create(int(world), int(rank), ctypes.c_char_p(unique_id), int(device), ctypes.byref(self._h)))
        self.world, self.rank, self.device = int(world), int(rank), int(device)

    @property
    def handle(self):
        return self._h

    def info(self):
        """(ncclCommCount, ncclCommUserRank) of the communicator."""
        c, r = ctypes.c_int(0), ctypes.c_int(0)
        _check(self._lib.tfft_dist_comm_info(self._h, ctypes.byref(c), ctypes.byref(r)))
        return int(c.value), int(r.value)

    def close(self):
        h = getattr(self, "_h", None)
        if h:
            self._h = None
            self._lib.tfft_dist_comm_destroy(h)
    # (no __del__: destroying a communicator from the garbage collector at interpreter shutdown, after the HIP runtime's own
    # teardown has begun, can hang; a communicator that is never closed is reclaimed with the process)


class DistPlan:
    """Owning wrapper of tfft_dist_plan: this rank's share of ONE transform spread over `world` GPUs (include/tfft.h).
    comm: a DistComm (the RCCL exchange then runs inside exec()) or None (pre() / post() only: the caller moves the
    chunks between them, over whatever transport it has; `buffers=` hands in its own send / receive tensors)."""

    def __init__(self, n, world, rank, device=0, comm=None, buffers=None, self_via_comm=False, slabs=1):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        self._comm = comm
        if slabs not in (1, 2, 4):
            raise TfftError(5, "slabs must be 1, 2 or 4")
        # caller-owned exchange buffers: the plan then allocates none of its own (TFFT_DIST_CALLER_BUFFERS; 512 MiB at 2^26 on one rank)
        # slabs > 1: the exchange overlaps the column pass slab by slab inside exec() (TFFT_DIST_SLABS_*)
        flags = ((DIST_SELF_VIA_COMM if self_via_comm else 0) | (DIST_CALLER_BUFFERS if buffers is not None else 0)
                 | {1: 0, 2: DIST_SLABS_2, 4: DIST_SLABS_4}[slabs])
        _check(self._lib.tfft_dist_plan_create(int(n), int(world), int(rank), int(device), comm.handle if comm else None,
                                               flags, ctypes.byref(self._h)))
        self.device = int(device)
        g = DistGeometry()
        _check(self._lib.tfft_dist_plan_geometry(self._h, ctypes.byref(g)))
        self.geometry = g
        self.local = int(g.n) // int(g.world)
        self._bufs = None
        if buffers is not None:
            self.set_buffers(*buffers)

    def set_buffers(self, send_re, send_im, recv_re, recv_im):
        import torch

        for t in (send_re, send_im, recv_re, recv_im):
            if not (t.is_cuda and t.dtype == torch.float16 and t.is_contiguous() and t.numel() >= self.local):
                raise TfftError(5, "exchange buffers must be contiguous CUDA float16 tensors of N / world elements")
        _check(self._lib.tfft_dist_plan_set_buffers(self._h, send_re.data_ptr(), send_im.data_ptr(), recv_re.data_ptr(),
                                                    recv_im.data_ptr()))
        self._bufs = (send_re, send_im, recv_re, recv_im)

    def _stream(self, stream):
        import torch

        return torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream

    def _planes(self, *tensors):
        import torch

        for t in tensors:
            if not (t.is_cuda and t.dtype == torch.float16 and t.is_contiguous() and t.numel() >= self.local
                    and t.device.index == self.device):
                raise TfftError(5, "planes must be contiguous CUDA float16 tensors of N / world elements on the plan's device")

    def pre(self, in_re, in_im, stream=None):
        import torch

        self._planes(in_re, in_im)
        with torch.cuda.device(self.device):
            _check(self._lib.tfft_dist_exec_pre(self._h, in_re.data_ptr(), in_im.data_ptr(), self._stream(stream)))

    def exchange(self, stream=None):
        import torch

        with torch.cuda.device(self.device):
            _check(self._lib.tfft_dist_exec_exchange(self._h, self._stream(stream)))

    def post(self, out_re, out_im, stream=None):
        import torch

        self._planes(out_re, out_im)
        with torch.cuda.device(self.device):
            _check(self._lib.tfft_dist_exec_post(self._h, out_re.data_ptr(), out_im.data_ptr(), self._stream(stream)))

    def exec(self, in_re, in_im, out_re, out_im, stream=None):
        import torch

        self._planes(in_re, in_im, out_re, out_im)
        with torch.cuda.device(self.device):
            _check(self._lib.tfft_dist_exec(self._h, in_re.data_ptr(), in_im.data_ptr(), out_re.data_ptr(), out_im.data_ptr(),
                                            self._stream(stream)))

    def close(self):
        h = getattr(self, "_h", None)
        if h:
            self._h = None
            self._lib.tfft_dist_plan_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001 - interpreter shutdown
            pass


def synth_uniform(re, im, n, batch, batch_stride=0, first_fft=0, seed=42, stream=None):
    """Fills planar CUDA half planes with the counter-hash uniform(-1, 1) signal (tfft_synth_uniform): element
    (first_fft + b, plane, j) is a pure function of the seed, so a CPU-side checker can regenerate any transform of it."""
    import torch

    stride = int(batch_stride) or 2 * int(n)
    need = (int(batch) - 1) * stride + int(n)
    for t in (re, im):
        if not (t.is_cuda and t.dtype == torch.float16 and t.is_contiguous() and t.numel() >= need):
            raise TfftError(5, "planes must be contiguous CUDA float16 tensors of (batch-1)*stride + n elements")
    dev = re.device.index
    if stream is None:
        stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        _check(load_library().tfft_synth_uniform(re.data_ptr(), im.data_ptr(), int(n), int(batch), stride, int(first_fft),
                                                 int(seed), stream))


def deinterleave(x_half2, out_re, out_im, stream=None):
    """(count, 2) interleaved half tensor -> planar re, im (tfft_deinterleave)."""
    import torch

    count = out_re.numel()
    if stream is None:
        stream = torch.cuda.current_stream(x_half2.device.index).cuda_stream
    with torch.cuda.device(x_half2.device.index):
        _check(load_library().tfft_deinterleave(x_half2.data_ptr(), out_re.data_ptr(), out_im.data_ptr(), count, stream))


def interleave(in_re, in_im, out_half2, stream=None):
    """planar re, im -> (count, 2) interleaved half tensor (tfft_interleave)."""
    import torch

    count = in_re.numel()
    if stream is None:
        stream = torch.cuda.current_stream(in_re.device.index).cuda_stream
    with torch.cuda.device(in_re.device.index):
        _check(load_library().tfft_interleave(in_re.data_ptr(), in_im.data_ptr(), out_half2.data_ptr(), count, stream))
