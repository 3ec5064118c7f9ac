"""One 1D transform spread over the GPUs of a node: four-step FFT with ONE exchange.

The reference has nothing to compare with (its multi-GPU code is commented out and ran independent FFTs
per device, src/base/ComputeFFT.h:295-411). MI355X-native plan: one process per GPU,
``torch.distributed`` over RCCL/xGMI. The exchange is one grouped send/recv (``batch_isend_irecv`` = one
ncclGroupStart/End) in which every rank sends one chunk of BOTH planes to each of its peers (all xGMI links busy at
once); everything either side of it is local.

N = N1 * N2, x viewed as [N1][N2] (n = n1 N2 + n2), X[k1 + N1 k2]:

    layout "columns" (input):   rank p owns x[n1 N2 + p C + c], c < C = N2 / P, stored [N1][C]
    1. FFT over n1 (length N1, strided axis, C columns innermost)            -> Y[k1][c]      local
    2. twiddle w_N^(k1 n2), n2 = p C + c                                                     local
    3. exchange: rows k1 in [q K, (q+1) K), K = N1 / P, go to rank q         (contiguous chunks, no packing)
    4. re-order [p'][k][c] -> [k][p' C + c]                                                   local
    5. FFT over n2 (length N2, contiguous rows, batch K)                     -> X[k1 + N1 k2]   local
    layout "transposed" (output): rank q owns k1 in its block, all k2, stored [K][N2]

Fused form (N1 = 256 or 512 and C a multiple of 64; this is what N = 2^26 on 8 GPUs takes): steps 1 and 2 are ONE
radix-N1 column pass whose epilogue applies the four-step twiddle (tfft_plan_opts.fourstep_n, column offset p C), so
step 4 is a pure re-order and at world size 1 disappears together with the exchange: the local work is then exactly
the library's transposed-order plan. General form (any other geometry): balanced split N1 ~ N2, steps 2 and 4 in
one fused re-order + twiddle kernel after the exchange.

``input_layout="natural"`` / ``output_layout="natural"`` (contiguous blocks of x / X per rank) cost one more
exchange each, plus a pack / unpack re-order.

With the GPU engine the fused form is a thin binding of the C ABI: ``tfft_dist_plan`` (include/tfft.h, csrc/dist.hpp) owns
the column pass, the exchange (``ncclSend`` / ``ncclRecv`` in one group on the caller's stream, through a communicator this
module creates with ``tfft_dist_comm_create`` from an id broadcast over the torch process group) and the row transforms,
which read the received chunks IN PLACE (segmented rows, no re-order pass) wherever N2 >= 2^16. ``transport="torch"``
keeps the exchange in Python (``batch_isend_irecv`` on the plan's send / receive tensors): what the tests use to run
several ranks on one GPU, and the hook for any other transport. A C++ host uses the same entry points through
``ComputeFFTMultiGPU`` of include/tensor_fft.hpp.

The general form, the natural layouts and every non-GPU engine keep the index logic in this class, with the arithmetic
delegated to an *engine* (:class:`HipEngine` = libtfft.so on the GPU). Tests drive that logic on CPU tensors over gloo with
an engine of their own.
"""


def _ilog2(x):
    return x.bit_length() - 1


class HipEngine:
    """Local arithmetic on the GPU through the C ABI (no other implementation exists in this package). Every
    intermediate lives in a buffer the engine allocates once per (role, size) and reuses on every forward()."""

    def __init__(self, device):
        from . import capi

        self.capi = capi
        self.device = device
        self._plans = {}
        self._bufs = {}

    def _plan(self, n, batch, inner, **kw):
        key = (n, batch, inner, tuple(sorted(kw.items())))
        p = self._plans.get(key)
        if p is None:
            import torch

            p = self.capi.TfftPlan(n, batch, self.device, inner=inner, in_batch_stride=n * inner,
                                   out_batch_stride=n * inner, preserve_input=True, **kw)
            if p.workspace_bytes:            # scratch owned by torch, handed over once
                p.set_workspace(torch.empty(p.workspace_bytes // 2, dtype=torch.float16, device=f"cuda:{self.device}"))
            self._plans[key] = p
        return p

    def buffers(self, role, like):
        """The (re, im) pair of this role, same shape / dtype / device as `like`; allocated on first use."""
        import torch

        key = (role, like.numel(), like.dtype, like.device)
        b = self._bufs.get(key)
        if b is None:
            b = (torch.empty_like(like), torch.empty_like(like))
            self._bufs[key] = b
        return b

    def fft_strided(self, re, im, n, inner):
        """FFT/n along axis 0 of [n][inner] planes."""
        o_re, o_im = self.buffers("strided", re)
        self._plan(n, 1, inner).exec(re, im, o_re, o_im)
        return o_re, o_im

    def supports_fourstep(self, n1, inner):
        return n1 in (256, 512) and inner >= 64 and inner % 64 == 0

    def fft_strided_fourstep(self, re, im, n1, inner, n_total, col0):
        """FFT/n1 along axis 0 of [n1][inner] planes, output row k of column c times w_{n_total}^(k (col0 + c)): one
        column pass (tfft_plan_opts.fourstep_n)."""
        o_re, o_im = self.buffers("strided", re)
        self._plan(n1, 1, inner, fourstep_n=n_total, fourstep_col0=col0).exec(re, im, o_re, o_im)
        return o_re, o_im

    def fft_rows(self, re, im, n, batch):
        """FFT/n of `batch` contiguous rows of length n."""
        o_re, o_im = self.buffers("rows", re)
        self._plan(n, batch, 1).exec(re, im, o_re, o_im)
        return o_re, o_im

    def permute_twiddle(self, re, im, a, b, c, n_tw=0, e0=0, role="permute"):
        """[a][b][c] -> [b][a][c], times w_n_tw^((e0 + b)(a c_total + c)) when n_tw > 0."""
        o_re, o_im = self.buffers(role, re)
        self.capi.permute_twiddle(re, im, o_re, o_im, a, b, c, n_tw, e0)
        return o_re, o_im

    def dist_geometry(self, n, world, rank):
        """The fused split the C ABI would choose (tfft_dist_geometry_query), or None when N is too small for it."""
        try:
            return self.capi.dist_geometry(n, world, rank)
        except self.capi.TfftError:
            return None

    def dist_plan(self, n, world, rank, comm=None, self_via_comm=False, slabs=1):
        """tfft_dist_plan for this rank with torch-owned exchange and output buffers. Returns (plan, send_re, send_im,
        recv_re, recv_im, out_re, out_im); at world size 1 the receive buffers are the send buffers (unless the own chunk
        is routed through the communicator, a test aid)."""
        import torch

        loc = n // world
        dev = f"cuda:{self.device}"
        mk = lambda: torch.empty(loc, dtype=torch.float16, device=dev)      # noqa: E731
        send_re, send_im = mk(), mk()
        recv_re, recv_im = (mk(), mk()) if (world > 1 or self_via_comm) else (send_re, send_im)
        plan = self.capi.DistPlan(n, world, rank, self.device, comm=comm, buffers=(send_re, send_im, recv_re, recv_im),
                                  self_via_comm=self_via_comm, slabs=slabs)
        return plan, send_re, send_im, recv_re, recv_im, mk(), mk()


class DistSetupError(RuntimeError):
    """Creating the native communicator or the distributed plan failed on at least one rank. Raised on EVERY rank of the
    group together (the ranks agree on the outcome before anyone leaves the constructor), so callers can fall back to another
    transport in lock-step instead of leaving peers blocked in a collective."""


class DistributedFFT1D:
    def __init__(self, n, group=None, engine=None, input_layout="columns", output_layout="transposed", fused=None,
                 transport=None, self_via_comm=False, slabs=1):
        """slabs (1, 2 or 4; "rccl" transport): the exchange overlaps the column pass slab by slab (TFFT_DIST_SLABS_*, include/tfft.h).
        transport (GPU engine, fused form, more than one rank): "rccl" = the C ABI runs the exchange itself over a
        communicator created through it (default when the process group's backend is nccl); "torch" = this class runs it
        over torch.distributed on the plan's buffers (default otherwise). self_via_comm (with "rccl"): the own chunk goes
        through ncclSend / ncclRecv too, so that a single GPU exercises the collective path (tests)."""
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
        self.engine = engine
        # fused form: N1 = 256 or 512 with at least 64 columns per rank (and the engine can do it)
        self.fused = False
        can = getattr(engine, "supports_fourstep", None)
        if fused is not False and can is not None:
            # prefer the split whose N2 the library transforms in the fewest passes: a single-kernel length (<= 2^15),
            # 4096 first of all; otherwise N1 = 256 (256-byte row segments in the column pass)
            order = (512, 256) if (lg - 9 == 12 or (lg - 8 > 15 >= lg - 9)) else (256, 512)
            for n1 in order:
                n2 = n // n1
                if n1 < n and n1 % p == 0 and n2 % p == 0 and can(n1, n2 // p):
                    self.fused, self.n1, self.n2 = True, n1, n2
                    break
        if fused and not self.fused:
            raise ValueError("the fused four-step form needs N1 = 256 or 512 and at least 64 columns per rank")
        if not self.fused:
            self.n1 = 1 << ((lg + 1) // 2)
            self.n2 = n // self.n1
        if self.n2 % p or self.n1 % p or (self.n2 // p) % 8:
            raise ValueError(f"N = {n} is too small for {p} ranks (needs N2/P >= 8 columns per rank)")
        self.c = self.n2 // p          # columns per rank (step 1)
        self.k = self.n1 // p          # rows per rank (step 5)
        if input_layout not in ("columns", "natural") or output_layout not in ("transposed", "natural"):
            raise ValueError("unknown layout")
        self.input_layout, self.output_layout = input_layout, output_layout
        self._recv = {}
        # fused form on the GPU engine: the core (column pass, exchange, row transforms) is one tfft_dist_plan
        self._core = None
        self.transport = None
        if self.fused and hasattr(engine, "dist_plan"):
            g = engine.dist_geometry(n, p, self.rank)
            if g is not None:
                assert (g.n1, g.n2, g.cols, g.rows) == (self.n1, self.n2, self.c, self.k), "C and Python geometry disagree"
                if transport is None:
                    transport = "rccl" if (p > 1 and dist.is_initialized() and dist.get_backend(group) == "nccl") else "torch"
                if transport not in ("rccl", "torch"):
                    raise ValueError("transport must be 'rccl' or 'torch'")
                self.transport = transport
                self._via = bool(self_via_comm) and transport == "rccl"
                self._comm = None
                # Everything below can fail on one rank only (dlopen of librccl, ncclCommInitRank, a hipMalloc); a rank that left
                # the constructor alone would strand its peers in the next collective. So: every step ends with an agreement over
                # the process group, and all ranks raise DistSetupError together.
                why = None
                try:
                    if transport == "rccl" and (p > 1 or self._via):
                        self._comm = self._native_comm(engine)
                except Exception as e:      # noqa: BLE001
                    why = f"{type(e).__name__}: {e}"
                why = self._agree(why, "creating the RCCL communicator")
                if why is None:
                    try:
                        self._core = engine.dist_plan(n, p, self.rank, comm=self._comm, self_via_comm=self._via, slabs=slabs)
                        self.geometry = self._core[0].geometry
                    except Exception as e:      # noqa: BLE001
                        why = f"{type(e).__name__}: {e}"
                    why = self._agree(why, "creating the distributed plan")
                if why is not None:
                    self.close()
                    raise DistSetupError(why)

    def _agree(self, why, what):
        """All ranks learn whether `what` worked everywhere: returns None if it did, else a message (the local failure, or
        that another rank failed). One small all-reduce on the process group's own device type."""
        dist = self.dist
        if self.world == 1 or not dist.is_initialized():
            return why
        import torch

        dev = f"cuda:{self.engine.device}" if dist.get_backend(self.group) == "nccl" else "cpu"
        ok = torch.tensor([0.0 if why else 1.0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if float(ok[0]) == 1.0:
            return None
        return why or f"{what} failed on another rank"

    def close(self):
        """Destroys the distributed plan and the communicator this object created (idempotent)."""
        core, self._core = getattr(self, "_core", None), None
        if core is not None:
            core[0].close()
        comm, self._comm = getattr(self, "_comm", None), None
        if comm is not None:
            comm.close()

    def _native_comm(self, engine):
        """An RCCL communicator of this process group's ranks, created through the C ABI: rank 0 makes the id
        (tfft_dist_unique_id = ncclGetUniqueId), the torch process group carries it, every rank joins
        (tfft_dist_comm_create = ncclCommInitRank)."""
        dist = self.dist
        box = [None]
        if self.rank == 0:
            # rank 0 may fail right here (librccl not loadable, a missing symbol): the peers are about to enter the broadcast,
            # so what travels is either the id or the error, and everybody raises or continues together
            try:
                box = [engine.capi.dist_unique_id()]
            except Exception as e:      # noqa: BLE001
                box = [("error", f"{type(e).__name__}: {e}")]
        if self.world > 1:
            src = 0 if self.group is None else dist.get_global_rank(self.group, 0)
            dist.broadcast_object_list(box, src=src, group=self.group)
        if isinstance(box[0], tuple):
            raise RuntimeError("rank 0 could not create the RCCL unique id: " + box[0][1])
        return engine.capi.DistComm(self.world, self.rank, box[0], engine.device)

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

    def _exchange(self, re, im, role, out=None):
        """All-to-all of both planes as ONE grouped operation: chunk q of each plane goes to rank q (one
        ncclGroupStart/End over RCCL: 2 (P - 1) sends and as many receives, every xGMI link busy at once). The receive
        buffers are `out` (the tfft_dist_plan's receive tensors) or allocated once per role."""
        p = self.world
        if p == 1:
            return re, im
        if out is None:
            key = (role, re.numel(), re.dtype, re.device)
            out = self._recv.get(key)
            if out is None:
                out = (re.new_empty(re.shape), im.new_empty(im.shape))
                self._recv[key] = out
        o_re, o_im = out
        chunk = re.numel() // p
        dist = self.dist
        ops = []
        for q in range(p):
            sl = slice(q * chunk, (q + 1) * chunk)
            if q == self.rank:
                o_re[sl].copy_(re[sl])
                o_im[sl].copy_(im[sl])
                continue
            peer = q if self.group is None else dist.get_global_rank(self.group, q)
            ops.append(dist.P2POp(dist.isend, re[sl], peer, self.group))
            ops.append(dist.P2POp(dist.isend, im[sl], peer, self.group))
            ops.append(dist.P2POp(dist.irecv, o_re[sl], peer, self.group))
            ops.append(dist.P2POp(dist.irecv, o_im[sl], peer, self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return o_re, o_im

    def forward(self, re, im):
        """re, im: this rank's N/P samples (flat float16 tensors in the input layout) -> its N/P outputs. The returned
        tensors are buffers owned by the engine / this object: copy them before the next forward() if they must live."""
        e, p = self.engine, self.world
        loc = self.n // p
        if re.numel() != loc or im.numel() != loc:
            raise ValueError("each rank passes N / world_size samples per plane")
        if self.input_layout == "natural" and p > 1:
            # rank holds rows n1 in its block, all n2: [R][P][C] -> chunks [P][R][C], exchange -> [N1][C]
            rows = self.n1 // p
            re, im = e.permute_twiddle(re, im, rows, p, self.c, role="pack_in")
            re, im = self._exchange(re, im, "in")
        if self._core is not None:
            # the whole plain path in the C ABI: column pass with the four-step twiddle -> exchange -> row transforms
            plan, send_re, send_im, recv_re, recv_im, out_re, out_im = self._core
            if self.transport == "rccl" and (p > 1 or self._via):
                plan.exec(re, im, out_re, out_im)
            else:
                plan.pre(re, im)
                self._exchange(send_re, send_im, "main", out=(recv_re, recv_im))
                plan.post(out_re, out_im)
            re, im = out_re, out_im
        elif self.fused:
            # 1 + 2. column transforms with the four-step twiddle in their epilogue
            re, im = e.fft_strided_fourstep(re, im, self.n1, self.c, self.n, self.rank * self.c)
            # 3. the one exchange of the plain path, 4. pure re-order (nothing to do on one rank)
            re, im = self._exchange(re, im, "main")
            if p > 1:
                re, im = e.permute_twiddle(re, im, p, self.k, self.c, role="unpack")
        else:
            re, im = e.fft_strided(re, im, self.n1, self.c)
            re, im = self._exchange(re, im, "main")
            # [p'][k][c] -> [k][p' C + c], twiddle w_N^((rank K + k)(p' C + c))
            re, im = e.permute_twiddle(re, im, p, self.k, self.c, self.n, self.rank * self.k, role="unpack")
        if self._core is None:
            # 5. row transforms
            re, im = e.fft_rows(re, im, self.n2, self.k)
        if self.output_layout == "natural" and p > 1:
            # rank holds [K][N2] = X[k1 + N1 k2]; natural block q wants k2 in its block (N2/P values), all k1:
            # [K][P][C] -> chunks [P][K][C], exchange -> [P'][K][C] = [k1][c] for its k2 block, then
            # [k1 = N1][C] -> [C][N1] to make k1 the fast index.
            re, im = e.permute_twiddle(re, im, self.k, p, self.c, role="pack_out")
            re, im = self._exchange(re, im, "out")
            re, im = self._transpose_last(re, im, self.n1, self.c)
        elif self.output_layout == "natural":
            re, im = self._transpose_last(re, im, self.n1, self.c)
        return re, im

    def phase_times(self, re, im, reps=10):
        """Mean milliseconds of the three phases of the fused GPU path (column pass / exchange / row transforms), from HIP
        events on the one stream all three are enqueued on: where a distributed transform spends its time. Plain layouts only.
        Returns {"pre_ms", "exchange_ms", "post_ms"}; the output buffers hold a valid transform afterwards."""
        import torch

        if self._core is None or self.input_layout != "columns" or self.output_layout != "transposed":
            raise ValueError("phase_times: needs the fused GPU path with the plain layouts")
        plan, send_re, send_im, recv_re, recv_im, out_re, out_im = self._core
        native = self.transport == "rccl" and (self.world > 1 or self._via)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
        for r in range(reps):
            ev[r][0].record()
            plan.pre(re, im)
            ev[r][1].record()
            if native:
                plan.exchange()
            else:
                self._exchange(send_re, send_im, "main", out=(recv_re, recv_im))
            ev[r][2].record()
            plan.post(out_re, out_im)
            ev[r][3].record()
        torch.cuda.synchronize()
        mean = lambda i: sum(e[i].elapsed_time(e[i + 1]) for e in ev) / reps      # noqa: E731
        return {"pre_ms": mean(0), "exchange_ms": mean(1), "post_ms": mean(2)}

    def _transpose_last(self, re, im, rows, cols):
        """[rows][cols] -> [cols][rows] (k1 becomes the fast index); pure data movement into buffers kept by this object."""
        key = ("transpose", re.numel(), re.dtype, re.device)
        out = self._recv.get(key)
        if out is None:
            out = (re.new_empty(re.shape), im.new_empty(im.shape))
            self._recv[key] = out
        for src, dst in ((re, out[0]), (im, out[1])):
            dst.view(cols, rows).copy_(src.view(rows, cols).t())
        return out
