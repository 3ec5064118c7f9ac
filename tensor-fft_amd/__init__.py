"""tensor-fft_amd: MI355X (gfx950) tensor-core FFT, host side.

Import name: ``tensor_fft_amd`` (see ``tensor_fft_amd.py`` at the repository
root; the directory carries the upstream project's hyphen).

Two layers, both thin:

* :mod:`.capi` — ctypes binding of the C ABI ``include/tfft.h`` in
  ``libtfft.so`` (hand-written HIP, built in-tree by ``__graft_entry__.build``).
  There is NO CPU or PyTorch fallback: if the library is missing or the device is
  not gfx950, calls raise.
* :mod:`.reference_api` — the reference's own host interface for this path
  (``CreatePlan``, ``PlanWorksOnDevice``, ``GetMaxNoOptInSharedMem``,
  ``DataHandler``, ``DataBatchHandler``, ``ComputeFFT``; reference
  src/base/Plan.h, DataHandler.h, ComputeFFT.h) with the same names, argument
  meaning and error behaviour, on top of :mod:`.capi`. PyTorch supplies device
  memory and streams only.
"""
from .capi import (TfftError, TfftPlan, TfftPlan2D, device_check, kernel_list, lib_path, load_library, plan_cache_policy,  # noqa: F401
                   plan_default_variant, plan_describe, ref_create_plan, synth_uniform, transposed_n2, tuning_add, tuning_clear,
                   tuning_load, tuning_query, variant_check)
from .reference_api import (  # noqa: F401
    ComputeFFT,
    CreatePlan,
    DataBatchHandler,
    DataHandler,
    GetMaxNoOptInSharedMem,
    Mode_256,
    Mode_4096,
    Plan,
    PlanWorksOnDevice,
)

__all__ = [
    "TfftError", "TfftPlan", "TfftPlan2D", "device_check", "lib_path", "load_library", "plan_cache_policy", "plan_default_variant", "plan_describe", "ref_create_plan",
    "synth_uniform", "transposed_n2", "variant_check", "kernel_list", "tuning_add", "tuning_clear", "tuning_load", "tuning_query",
    "ComputeFFT", "CreatePlan", "DataBatchHandler", "DataHandler", "GetMaxNoOptInSharedMem",
    "Mode_256", "Mode_4096", "Plan", "PlanWorksOnDevice",
]
