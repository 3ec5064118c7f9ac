"""The reference's host interface for the hot path, over the C ABI.

Same names, argument meaning and error behaviour as CPestka/Tensor-FFT:

* ``Plan`` / ``CreatePlan``            src/base/Plan.h:18-39, 77-194 (and the tuner-file overload :197-255)
* ``PlanWorksOnDevice``                src/base/Plan.h:257-296
* ``GetMaxNoOptInSharedMem``           src/base/Plan.h:298-303
* ``DataHandler`` / ``DataBatchHandler``  src/base/DataHandler.h:22-82, 86-166
* ``ComputeFFT`` (both overloads)      src/base/ComputeFFT.h:54-151, 162-293

Error convention of the reference is kept: functions that return
``std::optional<std::string>`` return ``None`` on success and the message
otherwise; ``CreatePlan`` prints its message and returns ``None`` on failure.
Device memory is a torch CUDA tensor (plumbing only); every transform runs in
libtfft.so.
"""
import numpy as np

from . import capi

Mode_256 = 0
Mode_4096 = 1


class Plan:
    """Field-for-field ``struct Plan<Integer>`` (src/base/Plan.h:18-39)."""

    __slots__ = (
        "fft_length_", "amount_of_r16_steps_", "amount_of_r2_steps_", "base_fft_mode_", "results_in_results_",
        "base_fft_warps_per_block_", "base_fft_blocksize_", "base_fft_gridsize_", "base_fft_shared_mem_in_bytes_",
        "r16_warps_per_block_", "r16_blocksize_", "r16_gridsize_", "r16_shared_mem_in_bytes_", "r2_blocksize_",
        "_exec_plans", "_variant", "_tuned",
    )

    def __init__(self, s):
        self.fft_length_ = int(s.fft_length)
        self.amount_of_r16_steps_ = s.amount_of_r16_steps
        self.amount_of_r2_steps_ = s.amount_of_r2_steps
        self.base_fft_mode_ = s.base_fft_mode
        self.results_in_results_ = bool(s.results_in_results)
        self.base_fft_warps_per_block_ = s.base_fft_warps_per_block
        self.base_fft_blocksize_ = s.base_fft_blocksize
        self.base_fft_gridsize_ = s.base_fft_gridsize
        self.base_fft_shared_mem_in_bytes_ = s.base_fft_shared_mem_in_bytes
        self.r16_warps_per_block_ = s.r16_warps_per_block
        self.r16_blocksize_ = s.r16_blocksize
        self.r16_gridsize_ = s.r16_gridsize
        self.r16_shared_mem_in_bytes_ = s.r16_shared_mem_in_bytes
        self.r2_blocksize_ = s.r2_blocksize
        self._exec_plans = {}
        self._variant = 0          # MI355X tuner knob (sixth column of a tuner file), 0 = default
        self._tuned = []           # (batch, variant, launch_iters) per tuner-file line of this length (columns 6 - 8)


def CreatePlan(fft_length, mode=Mode_256, base_fft_warps_per_block=8, r16_warps_per_block=8, r2_blocksize=256):
    """CreatePlan(fft_length, mode, base_wpb, r16_wpb, r2_blocksize) or CreatePlan(fft_length, tuner_file)."""
    if isinstance(mode, str):
        return _create_plan_from_file(fft_length, mode)
    rc, s, msg = capi.ref_create_plan(fft_length, mode, base_fft_warps_per_block, r16_warps_per_block, r2_blocksize)
    if msg:
        print(msg)
    if rc != capi.TFFT_OK:
        return None
    return Plan(s)


def _create_plan_from_file(fft_length, tuner_results_file):
    """Tuner-file overload (src/base/Plan.h:197-255): lines `N mode base_wpb r16_wpb r2_bs`."""
    try:
        f = open(tuner_results_file)
    except OSError:
        print("Error! Failed to open tuner file.")
        return None
    plan = None
    with f:
        for line in f:
            tok = line.split()
            if len(tok) < 5:
                continue
            if int(float(tok[0])) != fft_length:
                continue
            if plan is None:
                mode = Mode_256 if int(tok[1]) == 256 else Mode_4096
                plan = CreatePlan(fft_length, mode, int(tok[2]), int(tok[3]), int(tok[4]))
                if plan is None:
                    return None
            if len(tok) >= 6:      # tools/tuner.py appends: kernel variant [launch_iters [batch it was tuned at]]
                try:
                    variant = int(tok[5])
                    iters = int(tok[6]) if len(tok) >= 7 else 0
                    batch = int(tok[7]) if len(tok) >= 8 else 0
                    capi.variant_check(fft_length, 1, variant)     # unknown or WRONG-result bits: refuse the line
                    if not 0 <= iters <= 65535 or batch < 0:
                        raise ValueError("launch_iters outside 0 .. 65535")
                except (ValueError, capi.TfftError) as e:
                    print(f"Error! Tuner file holds an unusable kernel variant for this fft length: {e}")
                    return None
                plan._tuned.append((batch, variant, iters))
                if len(plan._tuned) == 1:
                    plan._variant = variant
            else:
                break              # a plain reference line: nothing more to read for this length
    if plan is not None:
        return plan
    print("Error! Tuner file didnt contain requested fft length.")
    return None


def PlanWorksOnDevice(my_plan, device_id):
    try:
        capi.device_check(device_id)
    except capi.TfftError as e:
        print(e.message)
        return False
    return True


def GetMaxNoOptInSharedMem(device_id):
    return capi.load_library().tfft_max_no_optin_shared_mem(int(device_id))


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("tensor_fft_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
    return torch


class DataHandler:
    """One block of 4*N halves: in_RE | in_IM | out_RE | out_IM (src/base/DataHandler.h:25-36)."""

    def __init__(self, fft_length, device=None):
        torch = _torch()
        self.fft_length_ = int(fft_length)
        self.device_ = torch.cuda.current_device() if device is None else int(device)
        self._err = None
        try:
            self.dptr_data_ = torch.empty(4 * self.fft_length_, dtype=torch.float16, device=f"cuda:{self.device_}")
        except RuntimeError as e:      # the reference prints and carries on (DataHandler.h:27-30)
            print(str(e))
            self._err = str(e)
            self.dptr_data_ = None
            return
        n = self.fft_length_
        self.dptr_input_RE_ = self.dptr_data_[0:n]
        self.dptr_input_IM_ = self.dptr_data_[n:2 * n]
        self.dptr_results_RE_ = self.dptr_data_[2 * n:3 * n]
        self.dptr_results_IM_ = self.dptr_data_[3 * n:4 * n]

    def PeakAtLastError(self):
        return self._err

    def CopyDataHostToDevice(self, data):
        """data: 2*N float16, [RE | IM] (DataHandler.h:45-53)."""
        torch = _torch()
        src = np.ascontiguousarray(data, dtype=np.float16).reshape(-1)
        if src.size != 2 * self.fft_length_:
            return "invalid argument"
        self.dptr_data_[: 2 * self.fft_length_].copy_(torch.from_numpy(src))
        return None

    def CopyResultsDeviceToHost(self, data, results_in_results):
        """Fills `data` (2*N float16) from the results half or the input half (DataHandler.h:55-70)."""
        n = self.fft_length_
        src = self.dptr_data_[2 * n:4 * n] if results_in_results else self.dptr_data_[0:2 * n]
        np.asarray(data).reshape(-1)[: 2 * n] = src.cpu().numpy()
        return None


class DataBatchHandler:
    """B*4*N halves: [fft0_RE|fft0_IM|fft1_RE|...] inputs then results (src/base/DataHandler.h:89-115)."""

    def __init__(self, fft_length, amount_of_ffts, device=None):
        torch = _torch()
        self.fft_length_ = int(fft_length)
        self.amount_of_ffts_ = int(amount_of_ffts)
        self.device_ = torch.cuda.current_device() if device is None else int(device)
        self._err = None
        n, b = self.fft_length_, self.amount_of_ffts_
        try:
            self.dptr_data_ = torch.empty(b * 4 * n, dtype=torch.float16, device=f"cuda:{self.device_}")
        except RuntimeError as e:
            print(str(e))
            self._err = str(e)
            self.dptr_data_ = None
            return
        self._inputs = self.dptr_data_[: 2 * n * b]
        self._results = self.dptr_data_[2 * n * b:]
        self.dptr_input_RE_ = [self._inputs[2 * i * n: 2 * i * n + n] for i in range(b)]
        self.dptr_input_IM_ = [self._inputs[2 * i * n + n: 2 * i * n + 2 * n] for i in range(b)]
        self.dptr_results_RE_ = [self._results[2 * i * n: 2 * i * n + n] for i in range(b)]
        self.dptr_results_IM_ = [self._results[2 * i * n + n: 2 * i * n + 2 * n] for i in range(b)]

    def PeakAtLastError(self):
        return self._err

    def CopyDataHostToDevice(self, data):
        torch = _torch()
        src = np.ascontiguousarray(data, dtype=np.float16).reshape(-1)
        if src.size != 2 * self.fft_length_ * self.amount_of_ffts_:
            return "invalid argument"
        self._inputs.copy_(torch.from_numpy(src))
        torch.cuda.synchronize(self.device_)          # DataHandler.h:132
        return None

    def CopyResultsDeviceToHost(self, data, results_in_results):
        src = self._results if results_in_results else self._inputs
        np.asarray(data).reshape(-1)[: src.numel()] = src.cpu().numpy()
        return None


def tuned_for_batch(tuned, batch, default_variant=0):
    """(variant, launch_iters) for a batch: the tuner-file line of this length whose batch is nearest on a log scale (lines
    without a batch column count as tuned at every batch, with the lowest priority among ties)."""
    import math

    best, best_d = (default_variant, 0), None
    for b, variant, iters in tuned:
        d = abs(math.log2(max(batch, 1)) - math.log2(b)) if b > 0 else 1e9
        if best_d is None or d < best_d:
            best, best_d = (variant, iters), d
    return best


def _exec_plan(fft_plan, batch, device):
    key = (batch, device)
    p = fft_plan._exec_plans.get(key)
    if p is None:
        variant, iters = tuned_for_batch(fft_plan._tuned, batch, fft_plan._variant)
        p = capi.TfftPlan(fft_plan.fft_length_, batch, device, variant=variant, launch_iters=iters)
        fft_plan._exec_plans[key] = p
    return p


def ComputeFFT(fft_plan, data, max_no_optin_shared_mem=32768):
    """ComputeFFT(plan, DataHandler | DataBatchHandler[, max_no_optin_shared_mem]).

    Leaves the spectrum where the reference would: in the results half if
    ``fft_plan.results_in_results_`` else in the input half (Plan.h:25-26,109-115);
    the input half may be clobbered (ComputeFFT.h:89-93). Returns None or the
    error string. The single-FFT form does not synchronise (ComputeFFT.h:49-53);
    the batch form ends with a device synchronise (ComputeFFT.h:286).
    """
    torch = _torch()
    n = fft_plan.fft_length_
    try:
        if isinstance(data, DataBatchHandler):
            if data.fft_length_ != n:
                return "invalid argument"
            p = _exec_plan(fft_plan, data.amount_of_ffts_, data.device_)
            ins = data._inputs
            outs = data._results if fft_plan.results_in_results_ else data._inputs
            p.exec(ins, ins[n:], outs, outs[n:])
            torch.cuda.synchronize(data.device_)
        else:
            if data.fft_length_ != n:
                return "invalid argument"
            p = _exec_plan(fft_plan, 1, data.device_)
            blk = data.dptr_data_
            out = blk[2 * n:] if fft_plan.results_in_results_ else blk
            p.exec(blk, blk[n:], out, out[n:])
    except capi.TfftError as e:
        return e.message
    return None
