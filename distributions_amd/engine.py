"""The batched row engine: many Gibbs row updates per launch.

`Gibbs` is the reference's pattern -- one PitmanYor driver, one slave mixture
per feature column, one MixtureIdTracker (examples/mixture/main.py:59-123,
213-248) -- over a table of rows resident in HBM.  `ShardedGibbs` runs it with
one process per GPU: rows are block-partitioned, every rank keeps a replica of
the sufficient statistics, and each sub-sweep ends with one all-reduce of the
integer statistic deltas (RCCL over xGMI through torch.distributed).
"""
import numpy as np

from . import _core

KIND = {"dd": _core.KIND_DD, "bb": _core.KIND_BB, "gp": _core.KIND_GP,
        "nich": _core.KIND_NICH, "dpd": _core.KIND_DPD,
        "bnb": _core.KIND_BNB}


def dd_shared(alphas):
    """DirichletDiscrete::Shared (models/dd.hpp:57-86)"""
    return _core.SharedParams.make(_core.KIND_DD, alphas=list(alphas))


def bb_shared(alpha, beta):
    """BetaBernoulli::Shared (models/bb.hpp:54-76)"""
    return _core.SharedParams.make(_core.KIND_BB, p=(alpha, beta))


def gp_shared(alpha, inv_beta):
    """GammaPoisson::Shared (models/gp.hpp:52-81)"""
    return _core.SharedParams.make(_core.KIND_GP, p=(alpha, inv_beta))


def nich_shared(mu, kappa, sigmasq, nu):
    """NormalInverseChiSq::Shared (models/nich.hpp:52-95)"""
    return _core.SharedParams.make(_core.KIND_NICH,
                                   p=(mu, kappa, sigmasq, nu))


def bnb_shared(alpha, beta, r):
    """BetaNegativeBinomial::Shared (models/bnb.hpp:50-84)"""
    return _core.SharedParams.make(_core.KIND_BNB,
                                   p=(alpha, beta, float(int(r))))


def dpd_shared(alpha, betas, beta0=0.0):
    """DirichletProcessDiscrete::Shared with values remapped to 0..len(betas)-1
    (models/dpd.hpp:59-153)"""
    return _core.SharedParams.make(_core.KIND_DPD, p=(alpha, beta0),
                                   betas=np.asarray(betas, np.float32))


class Gibbs(object):
    """Clustering driver + feature slaves over resident rows.  The clustering
    model is PitmanYor(alpha, d), or LowEntropy(dataset_size) when
    `dataset_size` is given (alpha and d are then ignored)."""

    def __init__(self, alpha, d, shareds, dataset_size=None):
        self.alpha = float(alpha)
        self.d = float(d)
        self.dataset_size = dataset_size
        self.shareds = list(shareds)
        self.core = _core.GibbsEngine(self.alpha, self.d, self.shareds,
                                      dataset_size)

    # -- data ---------------------------------------------------------------
    def load_rows(self, values, assign_packed, nonempty_groups,
                  empty_groups=1, row_offset=0):
        self.core.load_rows(values, assign_packed, nonempty_groups,
                            empty_groups, row_offset)

    def load_rows_torch(self, values, assign_packed, nonempty_groups,
                        empty_groups=1, row_offset=0):
        """values: int32/float32 CUDA tensors (one per feature);
        assign_packed: int32 CUDA tensor, rewritten in place to global ids."""
        ptrs = [int(v.data_ptr()) for v in values]
        self.core.load_rows_dev(ptrs, int(assign_packed.data_ptr()),
                                int(assign_packed.numel()), nonempty_groups,
                                empty_groups, row_offset,
                                keep=(list(values), assign_packed))

    def load_rows_unassigned(self, values, empty_groups=1, row_offset=0):
        """Rows without a group yet (a fresh mixture, examples/mixture/
        main.py:222-224); `init_sequential` assigns them."""
        self.core.load_rows_unassigned(values, empty_groups, row_offset)

    def init_sequential(self, row_begin, row_end, rng_state, prior_only=False):
        """The initialisation loop of examples/mixture/main.py on the
        device: rows [row_begin, row_end) are added one at a time -- score
        every group (all features, main.py:265-270; `prior_only`: the
        clustering model alone, main.py:227-232), sample, add.  Rows are
        assigned in order, starting at the first unassigned one.  Returns the
        new rng state."""
        return self.core.init_sequential(row_begin, row_end, rng_state,
                                         prior_only)

    # -- sweeps -------------------------------------------------------------
    def sweep(self, row_begin, row_end, batch_rows, seed, draw_base=0):
        """One pass over rows [row_begin,row_end) in frozen batches.

        Where the group set is normalised on the device (one feature with
        integer statistics: the default) the pass is QUEUED when this returns;
        the next sweep goes on without the host, and any other call on the
        engine (counts(), assignments(), ...) first waits for the device and
        pulls the host's copy of the group set.  `_core.synchronize()` waits
        explicitly."""
        self.core.sweep(row_begin, row_end, batch_rows, _core.rng_seed(seed),
                        draw_base)

    def sweep_sequential(self, row_begin, row_end, rng_state):
        """The reference's sequential chain; returns the new rng state."""
        return self.core.sweep_sequential(row_begin, row_end, rng_state)

    # -- state --------------------------------------------------------------
    def __len__(self):
        return self.core.group_count()

    def counts(self):
        return self.core.counts()

    def assignments(self):
        return self.core.assignments()

    def get_group(self, feature, groupid):
        return self.core.get_group(feature, groupid)

    def validate(self, raise_on_failure=True):
        """Mixture::validate for the whole engine (mixture.hpp:152-163,
        440-444): statistics against a recount from the rows, on the device
        (dist_gibbs_validate).  Returns the report; raises RuntimeError on
        the first inconsistency."""
        return self.core.validate(raise_on_failure)

    def row_scores(self, row):
        return self.core.row_scores(row)

    def kernel_stats(self, reset=False):
        return self.core.kernel_stats(reset)

    def set_option(self, name, value):
        self.core.set_option(name, value)

    def path_counts(self):
        return self.core.path_counts()


def sweep_sequential_many(engines, row_begin, row_end, rng_states):
    """M independent exact chains in ONE launch (BASELINE configs[3]:
    "independent chains"): engine i -- a `Gibbs` with its own rows -- runs the
    reference's sequential chain (examples/mixture/main.py:236-244) over its
    rows [row_begin, row_end) with rng_states[i]; one workgroup per chain,
    group creation and removal on the device.  The engines share one feature
    list.  -> the new rng states (numpy uint32)."""
    return _core.sweep_sequential_many([g.core for g in engines], row_begin,
                                       row_end, rng_states)


class ShardedGibbs(object):
    """Row-sharded Gibbs over the ranks of a torch.distributed process group.

    Every rank holds rows [row_offset, row_offset + n_local) and a full
    replica of the statistics.  A sub-sweep scores `batch_rows` local rows per
    rank against the common snapshot, turns the local moves into integer
    deltas, sums the deltas over ranks with ONE all-reduce, applies the sum
    and normalises the group set identically on every rank.  Row i always
    uses engine draw (draw_base + global index of i), so the result does not
    depend on the number of ranks for a given batch composition.

    `backend` is the per-rank compute object (a GibbsEngine-like); the
    distributed tests substitute a CPU stand-in to exercise this driver with
    gloo.
    """

    def __init__(self, backend, n_local, row_offset, group=None, device=None,
                 force_collective=False, columns=None, assign_packed=None):
        """columns: the local value columns (per feature a 4-byte-per-row
        tensor, as given to load_rows); assign_packed: the local initial
        assignment.  Both are needed only when a feature has order-dependent
        statistics (NormalInverseChiSq, GammaPoisson's log_prod): those are
        exchanged as rows, not as sums."""
        import torch.distributed as dist
        self.dist = dist
        self.backend = backend
        self.n_local = int(n_local)
        self.row_offset = int(row_offset)
        self.group = group
        self.device = device
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force_collective: take the delta + all-reduce path even with one
        # rank (what N > 1 runs, exercised on a single GPU)
        self.collective = self.world > 1 or (force_collective
                                             and dist.is_initialized())
        self._delta = None
        self._comm = None
        self._n_batches = {}    # batch_rows -> sub-sweeps per pass (all ranks)
        self.columns = columns
        self.assign_packed = assign_packed
        # (a property of the feature list: engines without such statistics
        # are never asked again -- the question would close an open
        # device-normalised run)
        self._has_ordered = backend.ordered_features() > 0
        self._exchange_mode()

    def _exchange_mode(self):
        """Order-dependent statistics travel as rows (every replica replays
        all of them) unless the backend keeps them as sums (float_stats = 1:
        one more all-reduce of a few doubles per group).  Asked again before
        every pass: whether the sums fit the backend's LDS pass depends on the
        group count, which grows."""
        backend = self.backend
        if not self._has_ordered:
            self.merged = self.ordered = False
            return
        self.merged = (self.collective
                       and getattr(backend, "float_delta_words",
                                   lambda: 0)() > 0)
        self.ordered = self.collective and not self.merged
        if self.ordered and self.columns is None:
            raise ValueError("features with order-dependent statistics need "
                             "the local value columns (columns=...)")

    def _gather_rows(self, arrays, n_pad):
        """arrays: list of (tensor-or-None, n valid) with 4-byte elements ->
        per array the concatenation over ranks (rank order) of the tensors
        padded to n_pad with 0xFFFFFFFF; ONE all-gather."""
        import torch
        pack = torch.full((len(arrays), n_pad), -1, dtype=torch.int32,
                          device=self.device)
        for i, t in enumerate(arrays):
            if t is not None and t.numel():
                pack[i, :t.numel()] = t.view(torch.int32)
        if self.world == 1:
            return [pack[i] for i in range(len(arrays))]
        if self._stage_through_host(pack):
            host = pack.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world)]
            self.dist.all_gather(parts, host, group=self.group)
            out = torch.stack(parts).to(pack.device)
        else:
            out = torch.empty((self.world,) + tuple(pack.shape),
                              dtype=torch.int32, device=self.device)
            self.dist.all_gather(list(out.unbind(0)), pack, group=self.group)
        return [out[:, i, :].reshape(-1).contiguous()
                for i in range(len(arrays))]

    def _replay(self, old, new, cols, n_pad, reset):
        """cols: per feature the local slice (or None)"""
        arrays = [new] + ([old] if old is not None else []) + [
            c for c in cols if c is not None]
        got = self._gather_rows(arrays, n_pad)
        new_all = got[0]
        old_all = got[1] if old is not None else None
        rest = got[2:] if old is not None else got[1:]
        ptrs, j = [], 0
        for c in cols:
            if c is None:
                ptrs.append(0)
            else:
                ptrs.append(int(rest[j].data_ptr()))
                j += 1
        self.backend.replay_ordered_dev(
            int(old_all.data_ptr()) if old_all is not None else 0,
            int(new_all.data_ptr()), ptrs, int(new_all.numel()), bool(reset))

    def _is_ordered(self, f):
        return self.backend.feature_is_ordered(f)

    def _stage_through_host(self, tensor):
        """device tensors under a host-only backend (gloo): the multi-rank
        tests on a single GPU run the real engine this way"""
        return tensor.is_cuda and self.dist.get_backend(self.group) == "gloo"

    def _all_reduce(self, tensor, op=None):
        if not self.collective:
            return
        op = op if op is not None else self.dist.ReduceOp.SUM
        if self._stage_through_host(tensor):
            host = tensor.cpu()
            self.dist.all_reduce(host, op=op, group=self.group)
            tensor.copy_(host)
        else:
            self.dist.all_reduce(tensor, op=op, group=self.group)

    def sync_initial_stats(self):
        """After every rank loaded ITS rows: make the statistics global."""
        import torch
        if not self.collective:
            return
        n = self.backend.stat_words()
        t = torch.zeros(n, dtype=torch.int32, device=self.device)
        self.backend.export_stats_dev(int(t.data_ptr()))
        if self.merged:   # (from the rank's OWN statistics: before the import)
            words = self.backend.float_delta_words()
            image = torch.zeros(words, dtype=torch.float64, device=self.device)
            self.backend.export_float_moments_dev(int(image.data_ptr()))
        self._all_reduce(t)
        self.backend.import_stats_dev(int(t.data_ptr()))
        if self.merged:
            self._all_reduce(image)
            self.backend.import_float_moments_dev(int(image.data_ptr()))
        if self.ordered:
            # order-dependent statistics: every replica replays ALL rows in
            # global order (Group::add_value per row, in row order)
            if self.assign_packed is None:
                raise ValueError("sync_initial_stats needs assign_packed")
            nmax = torch.tensor([self.n_local], dtype=torch.int64,
                                device=self.device)
            self._all_reduce(nmax, op=self.dist.ReduceOp.MAX)
            cols = [c if self._is_ordered(f) else None
                    for f, c in enumerate(self.columns)]
            self._replay(None, self.assign_packed, cols, int(nmax.item()),
                         reset=True)

    def use_native_comm(self, comm=None, transport=None):
        """Give the library its own communicator (or share `comm`, the
        `native_comm` of another engine of this process group): the sub-sweep
        loop then runs inside it, the all-reduce on the engine's stream (no
        Python and no stream hop per sub-sweep).  Collective.  transport:
        "rccl" (the default under an NCCL process group) or "host" -- the
        library's shared-memory transport for ranks that share one GPU (the
        default under gloo with device tensors: RCCL refuses two ranks on a
        device).  Returns False -- and the torch.distributed path stays in use
        -- when the engine has order-dependent statistics or the transport
        cannot be had."""
        import torch
        core = self.backend
        if not (self.collective and not self.ordered
                and hasattr(core, "sweep_sharded")):
            return False
        backend = self.dist.get_backend(self.group)
        if transport is None:
            transport = "rccl" if backend == "nccl" else "host"
        if transport == "rccl" and backend != "nccl":
            return False
        from . import _core
        if comm is not None:
            self._comm = comm
            return True
        # (flags and the id travel on whatever the process group moves:
        # device tensors under NCCL, host tensors under gloo)
        where = self.device if backend == "nccl" else "cpu"
        have = transport == "host" or _core.comm_available()
        ok = torch.tensor([1 if have else 0], dtype=torch.int32, device=where)
        self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN, group=self.group)
        if not int(ok.item()):
            return False
        rank = self.dist.get_rank(self.group)
        uid = torch.zeros(128, dtype=torch.uint8, device=where)
        if rank == 0:
            make = (_core.comm_unique_id_host if transport == "host"
                    else _core.comm_unique_id)
            uid.copy_(torch.from_numpy(make()))
        self.dist.broadcast(uid, src=self.dist.get_global_rank(
            self.group, 0) if self.group is not None else 0, group=self.group)
        try:
            comm = _core.Comm(uid.cpu().numpy(), rank, self.world)
        except RuntimeError:
            comm = None
        ok.fill_(0 if comm is None else 1)   # all ranks or none
        self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN, group=self.group)
        if not int(ok.item()):
            return False
        self._comm = comm
        return True

    def partition_by_value(self):
        """Collective, with a native communicator, after sync_initial_stats:
        the rows are placed so that no value of the (one, categorical)
        feature has rows on two ranks -- checked -- and the sub-sweeps
        exchange 3 words per group instead of the cells."""
        if self._comm is None:
            raise RuntimeError("partition_by_value needs use_native_comm()")
        self.backend.partition_by_value(self._comm)

    def gather_cells(self):
        """Collective: the replicas of value-partitioned ranks are whole again
        (before groups are read, validated, exported)."""
        self.backend.gather_cells(self._comm)

    @property
    def native_comm(self):
        return self._comm

    def sweep(self, batch_rows, seed_state, draw_base=0):
        """One pass over the local shard; all ranks take the same number of
        sub-sweeps (shards are equal up to one batch of padding)."""
        import torch
        self._exchange_mode()
        if self.ordered:
            self._comm = None   # (rows, not sums: the Python loop below)
        n_batches = (self.n_local + batch_rows - 1) // batch_rows
        if self.collective:
            if self._n_batches.get(batch_rows) is None:
                nb = torch.tensor([n_batches], dtype=torch.int64,
                                  device=self.device)
                self._all_reduce(nb, op=self.dist.ReduceOp.MAX)
                self._n_batches[batch_rows] = int(nb.item())
            n_batches = self._n_batches[batch_rows]
        if self._comm is not None:
            # (the library's loop; whether the group set is normalised on the
            # device is agreed among the ranks inside it, when a run opens)
            self.backend.sweep_sharded(self._comm, n_batches, batch_rows,
                                       seed_state, draw_base)
            return
        if not self.collective and hasattr(self.backend, "sweep"):
            # one rank, no exchange: the library's own loop (which queues the
            # whole pass without a host round trip where it can)
            self.backend.sweep(0, self.n_local, batch_rows, seed_state,
                               draw_base)
            return
        for b in range(n_batches):
            r0 = min(self.n_local, b * batch_rows)
            r1 = min(self.n_local, r0 + batch_rows)
            self.backend.batch_sample(r0, r1, seed_state, draw_base)
            if not self.collective:
                self.backend.batch_apply_local()
            else:
                n = self.backend.stat_words()
                if self._delta is None or self._delta.numel() < n:
                    # headroom: the image grows with the group count
                    self._delta = torch.empty(n + n // 4 + 1024,
                                              dtype=torch.int32,
                                              device=self.device)
                delta = self._delta[:n]
                self.backend.batch_delta_dev(int(delta.data_ptr()))
                self._all_reduce(delta)
                self.backend.batch_apply_delta_dev(int(delta.data_ptr()))
                if self.merged:
                    words = self.backend.float_delta_words()
                    fd = torch.empty(words, dtype=torch.float64,
                                     device=self.device)
                    self.backend.batch_float_delta_dev(int(fd.data_ptr()))
                    self._all_reduce(fd)
                    self.backend.batch_apply_float_delta_dev(
                        int(fd.data_ptr()))
                if self.ordered:
                    nb_rows = r1 - r0
                    old = torch.empty(nb_rows, dtype=torch.int32,
                                      device=self.device)
                    new = torch.empty(nb_rows, dtype=torch.int32,
                                      device=self.device)
                    self.backend.batch_moves_dev(int(old.data_ptr()),
                                                 int(new.data_ptr()))
                    cols = [c[r0:r1] if self._is_ordered(f) else None
                            for f, c in enumerate(self.columns)]
                    self._replay(old, new, cols, batch_rows, reset=False)
            self.backend.batch_finish()
