"""One 1D transform spread over the GPUs of a node: four-step FFT with ONE all-to-all.

The reference has nothing to compare with (its multi-GPU code is commented out and ran independent FFTs
per device, src/base/ComputeFFT.h:295-411). MI355X-native plan: one process per GPU,
``torch.distributed`` over RCCL/xGMI, the exchange is a single ``all_to_all_single`` per plane in which
every rank sends one chunk to each of its 7 peers (all xGMI links busy at once); everything either side of
it is local: radix passes along a strided axis, and one fused re-order + twiddle kernel.

N = N1 * N2, x viewed as [N1][N2] (n = n1 N2 + n2), X[k1 + N1 k2]:

    layout "columns" (input):   rank p owns x[n1 N2 + p C + c], c < C = N2 / P, stored [N1][C]
    1. FFT over n1 (length N1, strided axis, C columns innermost)            -> Y[k1][c]      local
    2. all-to-all: rows k1 in [q K, (q+1) K), K = N1 / P, go to rank q       (contiguous chunks, no packing)
    3. re-order [p'][k][c] -> [k][p' C + c] fused with the twiddle w_N^(k1 n2)                  local
    4. FFT over n2 (length N2, contiguous rows, batch K)                     -> X[k1 + N1 k2]   local
    layout "transposed" (output): rank q owns k1 in its block, all k2, stored [K][N2]

``input_layout="natural"`` / ``output_layout="natural"`` (contiguous blocks of x / X per rank) cost one more
all-to-all each, plus a pack / unpack re-order.

The class holds only index logic and the collective; arithmetic is delegated to an *engine*
(:class:`HipEngine` = libtfft.so on the GPU). Tests drive the same logic on CPU tensors over gloo with an
engine of their own.
"""
import math


def _ilog2(x):
    return x.bit_length() - 1


class HipEngine:
    """Local arithmetic on the GPU through the C ABI (no other implementation exists in this package)."""

    def __init__(self, device):
        from . import capi

        self.capi = capi
        self.device = device
        self._plans = {}

    def _plan(self, n, batch, inner):
        key = (n, batch, inner)
        p = self._plans.get(key)
        if p is None:
            p = self.capi.TfftPlan(n, batch, self.device, inner=inner, in_batch_stride=n * inner,
                                   out_batch_stride=n * inner, preserve_input=True)
            self._plans[key] = p
        return p

    def empty_like(self, t):
        import torch

        return torch.empty_like(t)

    def fft_strided(self, re, im, n, inner):
        """FFT/n along axis 0 of [n][inner] planes."""
        o_re, o_im = self.empty_like(re), self.empty_like(im)
        self._plan(n, 1, inner).exec(re, im, o_re, o_im)
        return o_re, o_im

    def fft_rows(self, re, im, n, batch):
        """FFT/n of `batch` contiguous rows of length n."""
        o_re, o_im = self.empty_like(re), self.empty_like(im)
        self._plan(n, batch, 1).exec(re, im, o_re, o_im)
        return o_re, o_im

    def permute_twiddle(self, re, im, a, b, c, n_tw=0, e0=0):
        """[a][b][c] -> [b][a][c], times w_n_tw^((e0 + b)(a c_total + c)) when n_tw > 0."""
        o_re, o_im = self.empty_like(re), self.empty_like(im)
        self.capi.permute_twiddle(re, im, o_re, o_im, a, b, c, n_tw, e0)
        return o_re, o_im


class DistributedFFT1D:
    def __init__(self, n, group=None, engine=None, input_layout="columns", output_layout="transposed"):
        import torch.distributed as dist

        if n & (n - 1) or n < 2:
            raise ValueError("Error! Input size has to be a power of 2!")
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        p = self.world
        if p & (p - 1):
            raise ValueError("the number of ranks has to be a power of 2")
        lg = _ilog2(n)
        self.n = n
        self.n1 = 1 << ((lg + 1) // 2)
        self.n2 = n // self.n1
        if self.n2 % p or self.n1 % p or (self.n2 // p) % 8:
            raise ValueError(f"N = {n} is too small for {p} ranks (needs N2/P >= 8 columns per rank)")
        self.c = self.n2 // p          # columns per rank (step 1)
        self.k = self.n1 // p          # rows per rank (step 4)
        if input_layout not in ("columns", "natural") or output_layout not in ("transposed", "natural"):
            raise ValueError("unknown layout")
        self.input_layout, self.output_layout = input_layout, output_layout
        self.engine = engine

    # ---- layouts (what each rank holds, as index arrays into x / X; used by callers and tests)
    def input_indices(self, rank=None):
        import numpy as np

        r = self.rank if rank is None else rank
        if self.input_layout == "natural":
            return np.arange(r * (self.n // self.world), (r + 1) * (self.n // self.world))
        n1 = np.arange(self.n1)[:, None]
        c = np.arange(self.c)[None, :]
        return (n1 * self.n2 + r * self.c + c).reshape(-1)

    def output_indices(self, rank=None):
        import numpy as np

        r = self.rank if rank is None else rank
        if self.output_layout == "natural":
            return np.arange(r * (self.n // self.world), (r + 1) * (self.n // self.world))
        k1 = r * self.k + np.arange(self.k)[:, None]
        k2 = np.arange(self.n2)[None, :]
        return (k1 + self.n1 * k2).reshape(-1)

    def _all_to_all(self, t):
        if self.world == 1:
            return t
        out = t.new_empty(t.shape)
        self.dist.all_to_all_single(out, t, group=self.group)
        return out

    def forward(self, re, im):
        """re, im: this rank's N/P samples (flat float16 tensors in the input layout) -> its N/P outputs."""
        e, p = self.engine, self.world
        loc = self.n // p
        if re.numel() != loc or im.numel() != loc:
            raise ValueError("each rank passes N / world_size samples per plane")
        if self.input_layout == "natural" and p > 1:
            # rank holds rows n1 in its block, all n2: [R][P][C] -> chunks [P][R][C], exchange -> [N1][C]
            rows = self.n1 // p
            re, im = e.permute_twiddle(re, im, rows, p, self.c)
            re, im = self._all_to_all(re), self._all_to_all(im)
        # 1. column transforms
        re, im = e.fft_strided(re, im, self.n1, self.c)
        # 2. the one exchange of the plain path
        re, im = self._all_to_all(re), self._all_to_all(im)
        # 3. [p'][k][c] -> [k][p' C + c], twiddle w_N^((rank K + k)(p' C + c))
        re, im = e.permute_twiddle(re, im, p, self.k, self.c, self.n, self.rank * self.k)
        # 4. row transforms
        re, im = e.fft_rows(re, im, self.n2, self.k)
        if self.output_layout == "natural" and p > 1:
            # rank holds [K][N2] = X[k1 + N1 k2]; natural block q wants k2 in its block (N2/P values), all k1:
            # [K][P][C] -> chunks [P][K][C], exchange -> [P'][K][C] = [k1][c] for its k2 block, then
            # [k1 = N1][C] -> [C][N1] to make k1 the fast index.
            re, im = e.permute_twiddle(re, im, self.k, p, self.c)
            re, im = self._all_to_all(re), self._all_to_all(im)
            re, im = self._transpose_last(re, im, self.n1, self.c)
        elif self.output_layout == "natural":
            re, im = self._transpose_last(re, im, self.n1, self.c)
        return re, im

    def _transpose_last(self, re, im, rows, cols):
        """[rows][cols] -> [cols][rows] (k1 becomes the fast index); pure data movement."""
        def tr(t):
            return t.reshape(rows, cols).t().contiguous().reshape(-1)

        return tr(re), tr(im)
