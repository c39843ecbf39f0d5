"""Gauss-ILRMA (IP) on ONE utterance with the frequency bins sharded over GPUs (SURVEY.md section 8, rows e / f2).

Utterance sharding (distributed.py) is the throughput mode; this is the latency mode for a single long utterance.
Every step of the reference's `GaussILRMA.update_once` (/root/reference/src/bss/ilrma.py:286-338) is independent per
bin except
  * the activation update, which reduces over f (ilrma.py:421-428)   -> one all-reduce of 2.N.K.T reals,
  * the power statistic mean_{f,t}|y_n|^2 (ilrma.py:304-307)           -> one all-reduce of N scalars,
  * the recorded loss (ilrma.py:648-677)                                -> one scalar, only when it is recorded.
A shard is a contiguous block of bins: X, W and the basis are sliced to it, the activation is replicated.  Each shard
runs the ordinary C-ABI entry points on its block plus the three split-update pieces of include/assx.h
(assx_ilrma_power_map, assx_nmf_half_sums, assx_nmf_apply_sums).

Determinism: the reductions are "gather the per-shard partials, add them in shard order" (never a ring/tree whose
association depends on the transport), and the shard count is part of the algorithm, not of the launch: `n_shards=S`
on one process gives bit-identical results to S ranks with one shard each (tests/test_distributed_gloo.py on CPU with
a NumPy stand-in for the shard ops, tests/test_gpu_multi.py with the HIP kernels).  Different S differ by rounding
only (summation order of the f-reduction).

Scope: algorithm_spatial in {'IP', 'ISS', 'IP2' / 'pairwise'} (the sweeps are per bin: ilrma.py:483-646), normalize in
{'power', 'projection-back', False} (projection back is per bin as well: ilrma.py:323-330, no exchange), n_basis <= 64,
no partitioning function (its latent variables couple all bins).
"""
import numpy as np
import torch
import torch.distributed as dist

from .. import _lib
from ..distributed import shard_range

EPS = 1e-12
THRESHOLD = 1e+12


class HipShardOps:
    """The per-shard steps on the HIP path.  All arrays carry a leading utterance axis of 1."""

    def __init__(self, dtype='float64', device=None):
        from ..ops import Engine
        self.eng = Engine(dtype=dtype, device=device)
        self.device = self.eng.dev
        self.real, self.cplx = self.eng.prec.real, self.eng.prec.cplx
        self._pb = {}

    def cov(self, X):
        B, M, F, T = X.shape
        return self.eng.cov_accumulate(X).reshape(B, F, M, M)

    def power_map(self, X, W):
        return self.eng.ilrma_power_map(X, W)

    def half_sums(self, half, P, Tb, V, domain, eps):
        # batch of the NMF entry point = sources: P (1,N,F,T) -> (N,F,T)
        return self.eng.nmf_half_sums(_lib.NMF_IS_MM, half, P[0], Tb[0], V[0], domain=domain, eps=eps)

    def apply_sums(self, A, sums, domain, eps):
        self.eng.nmf_apply_sums(_lib.NMF_IS_MM, A, sums, domain=domain, eps=eps)

    def spatial(self, X, W, Tb, V, C, domain, eps, threshold, status, spatial='IP', pair=(0, 1)):
        B, M, F, T = X.shape
        code = {'IP': _lib.SPATIAL_IP, 'ISS': _lib.SPATIAL_ISS, 'IP2': _lib.SPATIAL_IP2, 'pairwise': _lib.SPATIAL_IP2}[spatial]
        pb = None
        if C is not None:  # per-bin power statistic the entry point emits next to the sweep: one buffer per shard shape
            pb = self._pb.get((B, M, F))
            if pb is None:
                pb = self._pb[(B, M, F)] = self.eng.empty((B, M, F), dtype=torch.float64)
        self.eng.ilrma_spatial_update(X, W, Tb, V, domain=domain, eps=eps, threshold=threshold, status=status, C=C,
                                      power_bins=pb, spatial=code, pair=pair)

    def shard_power_mean(self, C, W, n_frames):
        """mean over the shard's bins of w_n^H C_f w_n: (N,) in the compute dtype (weighted by the shard's bin count
        when the shards are combined)."""
        return self.eng.power_from_cov(C, W, n_frames)[0]

    def ordered_sum(self, parts, weights=None):
        """(S, ...) -> (...): fixed ascending order, in a kernel (no torch arithmetic on the data path)."""
        w = None if weights is None else torch.tensor(weights, dtype=torch.float64, device=self.device)
        return self.eng.ordered_sum(parts, w)

    def normalize(self, W, Tb, power, domain, eps):
        self.eng.ilrma_normalize_power(W, Tb, power, domain=domain, eps=eps)

    def normalize_pb(self, X, W, Tb, ref, domain, status):
        """'projection-back' normalisation of one shard (ilrma.py:323-330): per bin, nothing to exchange."""
        scale = self.eng.projection_back_scale(X, W, ref, status)
        self.eng.ilrma_normalize_pb(W, Tb, scale, domain=domain)

    def loss(self, X, W, Tb, V, domain, eps):
        return self.eng.ilrma_loss(X, W, Tb, V, domain=domain, eps=eps)[0]

    def output(self, X, W, ref, status):
        scale = self.eng.projection_back_scale(X, W, ref, status)
        return self.eng.demix(X, W, scale=scale)

    def new_status(self):
        return self.eng.new_status(1)

    def check(self, status):
        if int(status.max().item()) & _lib.STATUS_SINGULAR:
            raise np.linalg.LinAlgError("Singular matrix")


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


class FrequencyShardedGaussILRMA:
    """
    Args (as GaussILRMA where they apply):
        n_shards: number of bin shards S (default: the world size).  Must be a multiple of the world size; rank r owns
            shards r*S/world ... (r+1)*S/world - 1.  Results depend on S (rounding), never on the world size.
        comm_device: where collective payloads live ("cpu" for a gloo group; default: the compute device).
        ops: the per-shard step implementation (default HipShardOps; tests inject a NumPy stand-in on CPU).
    """

    def __init__(self, n_basis=10, domain=2, normalize='power', reference_id=0, recordable_loss=True, eps=EPS,
                 threshold=THRESHOLD, *, algorithm_spatial='IP', dtype='float64', device=None, n_shards=None,
                 comm_device=None, ops=None):
        assert 1 <= domain <= 2, "1 <= `domain` <= 2 is not satisfied."
        if normalize not in ('power', 'projection-back', False):
            raise ValueError("Not support normalization based on {}. Choose 'power' or 'projection-back'".format(normalize))
        if algorithm_spatial not in ('IP', 'ISS', 'IP2', 'pairwise'):
            raise NotImplementedError("Not support {}-based spatial update.".format(algorithm_spatial))
        self.algorithm_spatial = algorithm_spatial
        self.update_pair = None
        if n_basis > 64:
            raise NotImplementedError("The F-sharded mode needs n_basis <= 64 (matrix-core NMF halves).")
        self.n_basis, self.domain, self.normalize = n_basis, domain, normalize
        self.reference_id, self.recordable_loss = reference_id, recordable_loss
        self.eps, self.threshold = eps, threshold
        self.loss = [] if recordable_loss else None
        self.rank, self.world = _world()
        self.n_shards = int(n_shards) if n_shards is not None else self.world
        if self.n_shards % self.world != 0:
            raise ValueError("n_shards ({}) must be a multiple of the world size ({})".format(self.n_shards, self.world))
        self.ops = ops if ops is not None else HipShardOps(dtype=dtype, device=device)
        self.comm_device = comm_device if comm_device is not None else self.ops.device

    # ------------------------------------------------------------------ deterministic reductions over shards
    def _ordered_sum(self, local_parts, weights=None):
        """Sum of one partial per shard (optionally weighted per shard), added in GLOBAL shard order on every rank."""
        t = torch.stack(local_parts).to(self.comm_device)
        if self.world > 1:
            parts = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(parts, t.contiguous())
            t = torch.cat(parts, dim=0)
        return self.ops.ordered_sum(t.to(self.ops.device), weights)

    # ------------------------------------------------------------------ driver
    def __call__(self, input, iteration=100, basis=None, activation=None):
        """input (n_channels, n_bins, n_frames) complex, the WHOLE utterance on every rank (NumPy or tensor; each rank
        keeps its bins).  basis (N,F,K) / activation (N,K,T): initial model; drawn from the global NumPy RNG in the
        reference's order on rank 0 and broadcast when omitted.  Returns the separated (N,F,T) complex128 array on
        every rank (NumPy)."""
        ops = self.ops
        X = input if isinstance(input, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(input))
        M, F, T = (int(v) for v in X.shape)
        N, K = M, self.n_basis
        self.n_sources = self.n_channels = M
        self.n_bins, self.n_frames = F, T
        self.update_pair = None
        if self.n_shards > F:
            raise ValueError("more shards ({}) than bins ({})".format(self.n_shards, F))
        if basis is None or activation is None:
            init = [None, None]
            if self.rank == 0:
                init = [np.random.rand(N, F, K) if basis is None else np.asarray(basis),
                        np.random.rand(N, K, T) if activation is None else np.asarray(activation)]
            if self.world > 1:
                dist.broadcast_object_list(init, src=0)
            basis, activation = init
        per_rank = self.n_shards // self.world
        my = list(range(self.rank * per_rank, (self.rank + 1) * per_rank))
        self._ranges = [shard_range(F, self.n_shards, s) for s in range(self.n_shards)]
        Xs, Ws, Ts, Cs = [], [], [], []
        for s in my:
            lo, hi = self._ranges[s]
            Xs.append(X[:, lo:hi, :].to(device=ops.device, dtype=ops.cplx).contiguous().unsqueeze(0))
            W = torch.eye(N, M, dtype=ops.cplx, device=ops.device).repeat(1, hi - lo, 1, 1).contiguous()
            Ws.append(W)
            Ts.append(torch.from_numpy(np.ascontiguousarray(np.asarray(basis)[:, lo:hi, :])).to(
                device=ops.device, dtype=ops.real).contiguous().unsqueeze(0))
            Cs.append(ops.cov(Xs[-1]) if self.normalize == 'power' else None)
        V = torch.from_numpy(np.ascontiguousarray(activation)).to(device=ops.device, dtype=ops.real).contiguous().unsqueeze(0)
        self._Xs, self._Ws, self._Ts, self._Cs, self._V, self._my = Xs, Ws, Ts, Cs, V, my
        self._status = ops.new_status()

        if self.recordable_loss:
            self.loss.append(self.compute_negative_loglikelihood())
        for _ in range(iteration):
            self.update_once()
            if self.recordable_loss:
                self.loss.append(self.compute_negative_loglikelihood())
        ops.check(self._status)

        # projection back is per bin: local, then the bins are gathered
        Ys = [ops.output(Xs[i], Ws[i], self.reference_id, self._status) for i in range(len(my))]
        ops.check(self._status)
        Y = self._gather_bins([y[0] for y in Ys], axis=1)
        self.estimation = Y.cpu().numpy().astype(np.complex128)
        return self.estimation

    def update_once(self):
        ops, d, eps = self.ops, self.domain, self.eps
        Xs, Ws, Ts, V = self._Xs, self._Ws, self._Ts, self._V
        pairwise = self.algorithm_spatial in ('IP2', 'pairwise')
        if pairwise:  # (0,1), (1,2), ..., (N-1,0)   (ilrma.py:635-646): host-side integer logic, the same on every rank
            m, n = (0, 1) if self.update_pair is None else ((self.update_pair[0] + 1) % self.n_sources,
                                                           (self.update_pair[1] + 1) % self.n_sources)
            self.update_pair = m, n

        def apply(A, sums):  # A (N, ...), sums (2, N, count): every source, or the selected pair only (ilrma.py:432-481)
            if not pairwise:
                ops.apply_sums(A, sums, d, eps)
            else:
                for src in self.update_pair:
                    ops.apply_sums(A[src:src + 1], sums[:, src:src + 1].contiguous(), d, eps)
        Ps = []
        # ---- source model: basis (per bin: local), then activation (reduce over f: the one real exchange)
        for i in range(len(Xs)):
            P = ops.power_map(Xs[i], Ws[i])
            apply(Ts[i][0], ops.half_sums(0, P, Ts[i], V, d, eps))
            Ps.append(P)
        act = self._ordered_sum([ops.half_sums(1, Ps[i], Ts[i], V, d, eps) for i in range(len(Xs))])
        apply(V[0], act)
        del Ps
        # ---- spatial model: covariance + the sweep (IP / ISS / IP2), per bin
        for i in range(len(Xs)):
            if self.algorithm_spatial == 'IP':
                ops.spatial(Xs[i], Ws[i], Ts[i], V, self._Cs[i], d, eps, self.threshold, self._status)
            else:
                ops.spatial(Xs[i], Ws[i], Ts[i], V, self._Cs[i], d, eps, self.threshold, self._status,
                            spatial=self.algorithm_spatial, pair=self.update_pair if pairwise else (0, 1))
        # ---- power normalisation: N scalars
        if self.normalize == 'power':
            # mean over all bins = sum_s (F_s / F) * mean over shard s
            wts = [(hi - lo) / float(self.n_bins) for lo, hi in self._ranges]
            power = self._ordered_sum([ops.shard_power_mean(self._Cs[i], Ws[i], self.n_frames) for i in range(len(Xs))],
                                      weights=wts).reshape(1, -1).contiguous()
            for i in range(len(Xs)):
                ops.normalize(Ws[i], Ts[i], power, d, eps)
        elif self.normalize == 'projection-back':
            for i in range(len(Xs)):
                ops.normalize_pb(Xs[i], Ws[i], Ts[i], self.reference_id, d, self._status)

    def compute_negative_loglikelihood(self):
        parts = [self.ops.loss(self._Xs[i], self._Ws[i], self._Ts[i], self._V, self.domain, self.eps).reshape(1)
                 for i in range(len(self._Xs))]
        return float(self._ordered_sum(parts).item())

    # ------------------------------------------------------------------ state, gathered over the shards
    def _gather_bins(self, local_blocks, axis):
        """Concatenate per-shard blocks along the bin axis in shard order, on every rank."""
        if self.world == 1:
            return torch.cat(local_blocks, dim=axis)
        blocks = [None] * self.n_shards
        per_rank = self.n_shards // self.world
        for j in range(per_rank):  # j-th local shard of every rank travels together; ragged -> pad to the widest
            widths = [self._ranges[r * per_rank + j][1] - self._ranges[r * per_rank + j][0] for r in range(self.world)]
            wmax = max(widths)
            b = local_blocks[j].movedim(axis, 0).contiguous()
            pad = torch.zeros((wmax,) + tuple(b.shape[1:]), dtype=b.dtype, device=self.comm_device)
            pad[: b.shape[0]] = b.to(self.comm_device)
            real = torch.view_as_real(pad) if pad.is_complex() else pad
            recv = [torch.empty_like(real) for _ in range(self.world)]
            dist.all_gather(recv, real.contiguous())
            for r in range(self.world):
                t = torch.view_as_complex(recv[r]) if pad.is_complex() else recv[r]
                blocks[r * per_rank + j] = t[: widths[r]].movedim(0, axis)
        return torch.cat(blocks, dim=axis)

    def gather_state(self):
        """COLLECTIVE: every rank must call it.  Gathers the bin-sharded model over the ranks and returns
        (demix_filter (F,N,M), basis (N,F,K), activation (N,K,T)) as NumPy arrays on every rank."""
        W = self._gather_bins([w[0] for w in self._Ws], axis=0).cpu().numpy().astype(np.complex128)
        Tb = self._gather_bins([t[0] for t in self._Ts], axis=1).cpu().numpy().astype(np.float64)
        return W, Tb, self.activation

    @property
    def demix_filter(self):
        """The gathered filter.  With more than one rank this is a COLLECTIVE (an all-gather over the bin shards): read it
        on every rank or on none -- `if rank == 0: model.demix_filter` hangs the job.  `gather_state()` says so by name."""
        return self._gather_bins([w[0] for w in self._Ws], axis=0).cpu().numpy().astype(np.complex128)

    @property
    def basis(self):
        """The gathered basis: a COLLECTIVE like `demix_filter`."""
        return self._gather_bins([t[0] for t in self._Ts], axis=1).cpu().numpy().astype(np.float64)

    @property
    def activation(self):
        return self._V[0].cpu().numpy().astype(np.float64)

    def __repr__(self):
        return "FrequencySharded-Gauss-ILRMA(n_basis={}, domain={}, normalize={}, algorithm_spatial={}, n_shards={})".format(
            self.n_basis, self.domain, self.normalize, self.algorithm_spatial, self.n_shards)
