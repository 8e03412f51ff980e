"""Host-side mirror of `latticenet.Lattice` / `latticenet.ModelParams` (un-vendored pybind module of the
reference, README.md:47) over the C ABI of libtln_hip.so.

Only the surface the reference touches is reproduced (call sites: train_ln.py:80,106,220,239;
seq_lattice/lattice_modules.py:38,64,284-304; seq_lattice/models.py:29-37,63-64,298,460).
"""
import ctypes as C

import torch

from . import _lib
from . import cfg as _cfg

__all__ = ["Lattice", "ModelParams", "HashTable", "stream_ptr"]


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """hipStream_t of torch's current stream.  torch.cuda.current_stream() builds a Python Stream object (~8 us,
    a quarter of the host time of a frame); the raw accessor is a plain C call."""
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class _DevView:
    """zero-copy torch view of handle-owned device memory via __cuda_array_interface__"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


class HashTable:
    """imported but unused by the reference (lattice_modules.py:7); kept so the import resolves"""


class Lattice:
    """One level of the permutohedral lattice.  Level 0 owns the native handle; coarser levels are children
    of it and live as long as it does (append-only numbering at every level is what the temporal fusion
    modules rely on, lattice_modules.py:59-60, 213-215)."""

    def __init__(self, handle, sigmas, capacity, root=None, name="lattice"):
        self._h = handle
        self._root = root            # keeps level 0 (and thus the native memory) alive
        self._owner = root is None
        self._sigmas = list(sigmas)
        self._capacity = int(capacity)
        self._values = None
        self._coarse = None
        self._csr_key = None         # (indices tensor, _version) the native CSR was built from
        self.name = name

    # ---- construction -----------------------------------------------------------------
    @staticmethod
    def create(config_file, name="lattice"):
        """Lattice.create(cfg, "lattice") of the reference (train_ln.py:106): reads `lattice_gpu`."""
        lg = _cfg.load(config_file)["lattice_gpu"]
        nr_sigmas = int(lg["nr_sigmas"])
        sigmas = []
        for i in range(nr_sigmas):
            val, extent = str(lg["sigma_%d" % i]).split()
            sigmas += [float(val)] * int(extent)
        return Lattice.from_params(sigmas, int(lg["hash_table_capacity"]), name,
                                   scale_constant=lg.get("scale_constant", None))

    @staticmethod
    def parse_scale_constant(v):
        """cfg key lattice_gpu.scale_constant (not in the reference's cfg: the constant is hard-coded in its un-vendored
        lattice_net dependency, README.md:47): None / "adams" -> (d+1)*sqrt(2/3), the default;
        "unit" -> 1.0 (the factor dropped); or a positive number.  UNVERIFIED against upstream until a real checkpoint
        or a scan with known vertex counts is at hand; cfg:71's sizing hint is what speaks for the default."""
        if v is None:
            return 0.0
        if isinstance(v, str):
            t = v.strip().lower()
            if t in ("", "adams", "default"):
                return 0.0
            if t in ("unit", "one", "none"):
                return 1.0
            v = float(t)
        v = float(v)
        if not v > 0.0:
            raise _lib.TlnError("lattice_gpu.scale_constant must be positive, 'adams' or 'unit'")
        return v

    @staticmethod
    def from_params(sigmas, capacity, name="lattice", scale_constant=None):
        if not torch.cuda.is_available():
            raise _lib.TlnError("Lattice needs a HIP device (torch.cuda.is_available() is False)")
        sigmas = [float(s) for s in sigmas]
        if len(sigmas) != 3:
            raise _lib.TlnError("only pos_dim == 3 is supported, got %d sigmas" % len(sigmas))
        torch.cuda.current_stream()  # make sure the HIP context of the current device exists
        h = C.c_void_p()
        arr = (C.c_double * 3)(*sigmas)
        _lib.check(_lib.lib().tln_lattice_create_ex(C.byref(h), 3, arr, int(capacity),
                                                    Lattice.parse_scale_constant(scale_constant)), "tln_lattice_create_ex")
        return Lattice(h, sigmas, capacity, None, name)

    def __del__(self):
        try:
            if self._owner and self._h:
                _lib.lib().tln_lattice_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def memory_bytes(self):
        """device memory of the whole level stack by purpose (tln_lattice_memory)"""
        out = (C.c_int64 * 5)()
        _lib.check(_lib.lib().tln_lattice_memory(self._h, out), "tln_lattice_memory")
        return dict(zip(("structure", "row_workspaces", "pool_accumulators", "tables", "total"), (int(x) for x in out)))

    def apply_options(self):
        """hands the kernel-selection options in force on this host thread (options.py; none = the library's defaults) to
        the native handle (K1 variant / bucket size, pool kernel): the library keeps no process-wide switch"""
        from . import options as O
        g = O.generation()
        if getattr(self, "_opt_gen", 0) != g:
            _lib.check(_lib.lib().tln_lattice_set_options(self._h, O.current_ref()), "tln_lattice_set_options")
            self._opt_gen = g

    # ---- the API the reference uses -----------------------------------------------------
    def set_values(self, t):
        self._values = t            # aliasing, not copying (lm:284 then lm:290)

    def values(self):
        return self._values

    def val_dim(self):
        return int(self._values.shape[1]) if self._values is not None else 0

    def pos_dim(self):
        return 3

    def get_filter_extent(self, neighbourhood_size):
        return 2 * (self.pos_dim() + 1) * int(neighbourhood_size) + 1

    def nr_lattice_vertices(self):
        return int(_lib.lib().tln_lattice_nr_vertices(self._h))

    def capacity(self):
        return self._capacity

    def lvl(self):
        return int(_lib.lib().tln_lattice_level(self._h))

    def sigmas(self):
        return list(self._sigmas)

    def scale_constant(self):
        """c of scale_i = c / (sigma_i sqrt((i+1)(i+2))) this lattice (and its coarse levels) was created with"""
        return float(_lib.lib().tln_lattice_scale_constant(self._h))

    def bucket_fallbacks(self):
        """frames this lattice redid with the per-row-atomic K1 kernels because a bucket's LDS table overflowed"""
        return int(_lib.lib().tln_lattice_bucket_fallbacks(self._h))

    def overflow_rows(self):
        return int(_lib.lib().tln_lattice_overflow_rows(self._h))

    def clear(self):
        _lib.check(_lib.lib().tln_lattice_clear(self._h, stream_ptr()), "tln_lattice_clear")
        self._csr_key = None
        self.__dict__.pop("_tables", None)
        c = self._coarse
        while c is not None:
            c._csr_key = None
            c._values = None
            c.__dict__.pop("_tables", None)
            c = c._coarse

    # ---- structure ------------------------------------------------------------------------
    def keys(self):
        v = self.nr_lattice_vertices()
        out = torch.empty((v, 3), dtype=torch.int32, device="cuda")
        if v:
            _lib.check(_lib.lib().tln_lattice_keys(self._h, _ptr(out), v, stream_ptr()), "tln_lattice_keys")
        return out

    def insert_keys(self, keys):
        keys = keys.contiguous()
        out = torch.empty((keys.shape[0],), dtype=torch.int32, device="cuda")
        _lib.check(_lib.lib().tln_lattice_insert_keys(self._h, _ptr(keys), keys.shape[0], _ptr(out), stream_ptr()),
                   "tln_lattice_insert_keys")
        self._csr_key = None
        return out

    def _table(self, fn, rows):
        p = C.c_void_p()
        _lib.check(fn(self._h, C.byref(p), stream_ptr()), "neighbour table")
        return p, rows

    def neighbour_table_ptr(self):
        return self._table(_lib.lib().tln_neighbour_table, self.nr_lattice_vertices())[0]

    def coarse_to_fine_table_ptr(self):
        return self._table(_lib.lib().tln_coarse_to_fine_table, self.nr_lattice_vertices())[0]

    def fine_to_coarse_table_ptr(self):
        """valid on a COARSE level: [V_fine, 9] rows into this level"""
        return self._table(_lib.lib().tln_fine_to_coarse_table, 0)[0]

    @staticmethod
    def _table_tensor(ptr, rows):
        if rows == 0:
            return torch.empty((0, 9), dtype=torch.int32, device="cuda")
        return torch.as_tensor(_DevView(ptr.value, (rows, 9), "<i4"), device="cuda").clone()

    def _cached_table(self, name, key, make):
        """torch copies of the native tables, kept while the levels do not grow (used by the backward passes)"""
        cache = self.__dict__.setdefault("_tables", {})
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            hit = (key, make())
            cache[name] = hit
        return hit[1]

    def neighbour_table(self):
        v = self.nr_lattice_vertices()
        return self._cached_table("nbr", v, lambda: self._table_tensor(self.neighbour_table_ptr(), v))

    def coarse_to_fine_table(self):
        v = self.nr_lattice_vertices()
        vf = self._parent_vertices()
        return self._cached_table("c2f", (v, vf), lambda: self._table_tensor(self.coarse_to_fine_table_ptr(), v))

    def fine_to_coarse_table(self, nr_fine):
        v = self.nr_lattice_vertices()
        return self._cached_table("f2c", (v, nr_fine),
                                  lambda: self._table_tensor(self.fine_to_coarse_table_ptr(), nr_fine))

    def _parent_vertices(self):
        return self._parent.nr_lattice_vertices() if getattr(self, "_parent", None) is not None else -1

    def prepare_levels(self, nr_coarse_levels):
        """after distribute(): extend the coarse levels and build all their tables in as few launches as possible"""
        _lib.check(_lib.lib().tln_lattice_prepare_levels(self._h, int(nr_coarse_levels), stream_ptr()),
                   "tln_lattice_prepare_levels")

    def coarsen(self):
        """The persistent coarse level, extended by the vertices added since the last call."""
        p = C.c_void_p()
        _lib.check(_lib.lib().tln_coarsen(self._h, C.byref(p), stream_ptr()), "tln_coarsen")
        if self._coarse is None:
            self._coarse = Lattice(p, [2 * s for s in self._sigmas], self._capacity,
                                   self._root if self._root is not None else self, self.name + "_c")
            self._coarse._parent = self
        return self._coarse

    # ---- K1 -----------------------------------------------------------------------------
    def distribute(self, positions, values, reset_hashmap=True, subtract_mean=True):
        positions = positions.contiguous().float()
        n = positions.shape[0]
        if positions.dim() != 2 or positions.shape[1] != 3:
            raise _lib.TlnError("positions must be [N,3], got %s" % (tuple(positions.shape),))
        if values is None or values.numel() == 0:
            values, val_dim = None, 0
        else:
            values = values.contiguous().float()
            if values.shape[0] != n:
                raise _lib.TlnError("positions/values row mismatch")
            val_dim = values.shape[1]
        if reset_hashmap:
            self.clear()
        cols = 3 + val_dim + 1
        distributed = torch.empty((4 * n, cols), dtype=torch.float32, device="cuda")
        indices = torch.empty((4 * n,), dtype=torch.int32, device="cuda")
        weights = torch.empty((4 * n,), dtype=torch.float32, device="cuda")
        from . import ops as _ops
        self.apply_options()
        with _ops._timed("distribute", n=n):
            rc = _lib.lib().tln_distribute(self._h, _ptr(positions), _ptr(values), n, val_dim,
                                           1 if subtract_mean else 0, _ptr(distributed), _ptr(indices),
                                           _ptr(weights), stream_ptr())
        _lib.check(rc, "tln_distribute")
        # the handle now holds the frame's rows grouped by vertex ("bins"): the PointNet pool of these very tensors
        # reads them instead of a sorted row list; the sorted CSR is only built on demand (ensure_csr)
        self._csr_key = None
        self._bins_key = (distributed, distributed._version, indices, indices._version)
        self._last_indices = indices
        return distributed, indices, weights

    @staticmethod
    def distribute_batch(lattices, positions, values, reset_hashmap=True, subtract_mean=True):
        """distribute() for 1..8 lattices of lock-stepped sequences with ONE batch of K1 launches
        (tln_distribute_begin_multi: blockIdx.y = lattice) -> [(distributed, indices, weights), ...], bit for bit what
        the single calls return"""
        n = len(lattices)
        pos = [p.contiguous().float() for p in positions]
        vals = [v.contiguous().float() for v in values]
        val_dim = vals[0].shape[1]
        if reset_hashmap:
            hs = (C.c_void_p * n)(*[l._h for l in lattices])
            _lib.check(_lib.lib().tln_lattice_clear_multi(hs, n, stream_ptr()), "tln_lattice_clear_multi")
        for l in lattices:
            l.apply_options()
        calls = (_lib.DistributeCall * n)()
        outs = []
        for k, (l, p, v) in enumerate(zip(lattices, pos, vals)):
            rows = 4 * p.shape[0]
            d = torch.empty((rows, 3 + val_dim + 1), dtype=torch.float32, device="cuda")
            i = torch.empty((rows,), dtype=torch.int32, device="cuda")
            w = torch.empty((rows,), dtype=torch.float32, device="cuda")
            c = calls[k]
            c.l, c.d_positions, c.d_values, c.n, c.val_dim = l._h, _ptr(p), _ptr(v), p.shape[0], val_dim
            c.subtract_mean, c.d_distributed, c.d_indices, c.d_weights = 1 if subtract_mean else 0, _ptr(d), _ptr(i), _ptr(w)
            outs.append((d, i, w))
        _lib.check(_lib.lib().tln_distribute_begin_multi(calls, n, stream_ptr()), "tln_distribute_begin_multi")
        for l, (d, i, w) in zip(lattices, outs):
            _lib.check(_lib.lib().tln_distribute_finish(l._h, stream_ptr()), "tln_distribute_finish")
            l._csr_key = None
            l._bins_key = (d, d._version, i, i._version)
            l._last_indices = i
        return outs

    def bins_valid_for(self, distributed):
        k = getattr(self, "_bins_key", None)
        return k is not None and k[0].data_ptr() == distributed.data_ptr()

    def drop_bins(self):
        """the rows were edited after the distribute: the native pool must not take them from the bins"""
        _lib.check(_lib.lib().tln_lattice_drop_bins(self._h), "tln_lattice_drop_bins")
        self._bins_key = None

    def bins_describe(self, distributed, indices):
        """True when `distributed` / `indices` are the untouched outputs of the last distribute on this lattice"""
        k = getattr(self, "_bins_key", None)
        return k is not None and k[0] is distributed and k[1] == distributed._version and k[2] is indices and \
            k[3] == indices._version

    def csr(self):
        """(order, sorted_vertex, seg_start) of the native CSR left by the last distribute / ensure_csr: row ids
        sorted stably by vertex, rejected rows in the tail bucket V.  Test / debug read-out."""
        import ctypes
        if self._csr_key is None and getattr(self, "_last_indices", None) is not None:
            self.ensure_csr(self._last_indices)        # a distribute leaves bins, not the sorted list: build it now
        rows = ctypes.c_int64(0)
        _lib.check(_lib.lib().tln_lattice_csr(self._h, None, None, None, ctypes.byref(rows), stream_ptr()),
                   "tln_lattice_csr")
        r = rows.value
        order = torch.empty((r,), dtype=torch.int32, device="cuda")
        sv = torch.empty((r,), dtype=torch.int32, device="cuda")
        seg = torch.empty((self.nr_lattice_vertices() + 2,), dtype=torch.int32, device="cuda")
        if r:
            _lib.check(_lib.lib().tln_lattice_csr(self._h, _ptr(order), _ptr(sv), _ptr(seg), ctypes.byref(rows),
                                                  stream_ptr()), "tln_lattice_csr")
        return order, sv, seg

    def ensure_csr(self, indices):
        k = self._csr_key
        if k is not None and k[0] is indices and k[1] == indices._version:
            return
        idx = indices.contiguous().to(torch.int32)
        _lib.check(_lib.lib().tln_build_csr(self._h, _ptr(idx), idx.shape[0], stream_ptr()), "tln_build_csr")
        self._csr_key = (indices, indices._version)


class ModelParams:
    """latticenet.ModelParams (reference getters used at models.py:29-37, 63-64, 488-524): the `model`
    section of the cfg."""

    def __init__(self, d):
        self._d = dict(d)

    @staticmethod
    def create(config_file):
        return ModelParams(_cfg.load(config_file)["model"])

    def _get(self, k, default=None):
        return self._d.get(k, default)

    def positions_mode(self): return self._get("positions_mode", "xyz")
    def values_mode(self): return self._get("values_mode", "none")
    def pointnet_layers(self): return list(self._get("pointnet_layers", [16, 32, 64]))
    def pointnet_start_nr_channels(self): return int(self._get("pointnet_start_nr_channels", 64))
    def nr_downsamples(self): return int(self._get("nr_downsamples", 2))
    def nr_blocks_down_stage(self): return list(self._get("nr_blocks_down_stage", [2, 2, 2]))
    def nr_blocks_bottleneck(self): return int(self._get("nr_blocks_bottleneck", 3))
    def nr_blocks_up_stage(self): return list(self._get("nr_blocks_up_stage", [1, 2, 2]))
    def nr_levels_down_with_normal_resnet(self): return int(self._get("nr_levels_down_with_normal_resnet", 3))
    def nr_levels_up_with_normal_resnet(self): return int(self._get("nr_levels_up_with_normal_resnet", 3))
    def compression_factor(self): return float(self._get("compression_factor", 1.0))
    def dropout_last_layer(self): return float(self._get("dropout_last_layer", 0.0))
    def experiment(self): return self._get("experiment", "none")
