"""Host-side mirror of the reference's mesh/state containers (t_mesh, o_ARRAYS) over the C ABI.

`Mesh.load(meshdir, ...)` calls the C++ host mesh layer (csrc/mesh_host.cpp, a
restatement of mesh_setup + the ALE part of ocean_setup) and exposes every array as
a numpy view whose C shape is the reversed Fortran shape, e.g. Fortran
`area(nl, node)` -> numpy `(node, nl)` with the level index fastest in memory.
"""
import ctypes as C
import numpy as np
from . import _lib

WHICH_ALE = {"linfs": 0, "zlevel": 1, "zstar": 2}


def _shapes(d):
    N = d.myDim_nod2D + d.eDim_nod2D
    E = d.myDim_elem2D + d.eDim_elem2D
    EX = E + d.eXDim_elem2D
    myE = d.myDim_elem2D
    D = d.myDim_edge2D + d.eDim_edge2D
    myD = d.myDim_edge2D
    nl = d.nl
    return dict(
        myList_nod2D=(N,), myList_elem2D=(EX,), myList_edge2D=(D,), coord_nod2D=(N, 2), geo_coord_nod2D=(N, 2),
        elem2D_nodes=(EX, 3), edges=(D, 2), edge_tri=(D, 2), elem_edges=(myE, 3), elem_neighbors=(myE, 3),
        nod_in_elem2D=(N, d.max_nod_in_elem), nod_in_elem2D_num=(N,), nlevels=(EX,), ulevels=(EX,),
        nlevels_nod2D=(N,), ulevels_nod2D=(N,), nlevels_nod2D_min=(N,), ulevels_nod2D_max=(N,),
        zbar=(nl,), Z=(nl - 1,), depth=(N,), elem_area=(EX,), area=(N, nl), area_inv=(N, nl), areasvol=(N, nl),
        areasvol_inv=(N, nl), mesh_resolution=(N,), gradient_sca=(myE, 6), gradient_vec=(myE, 6),
        edge_dxdy=(D, 2), edge_cross_dxdy=(D, 4), elem_cos=(EX,), metric_factor=(EX,), coriolis=(myE,),
        coriolis_node=(N,), ssh_rowptr=(d.myDim_nod2D + 1,), ssh_colind=(d.ssh_nza,), ssh_colind_loc=(d.ssh_nza,),
        ssh_values=(d.ssh_nza,), edge_up_dn_tri=(myD, 2), zbar_n_bot=(N,), zbar_n_srf=(N,),
        bottom_node_thickness=(N,), zbar_e_bot=(E,), zbar_e_srf=(E,), bottom_elem_thickness=(myE,))


def state_shapes(d, ntr):
    N = d.myDim_nod2D + d.eDim_nod2D
    E = d.myDim_elem2D + d.eDim_elem2D
    myE = d.myDim_elem2D
    nl = d.nl
    return dict(tr_arr=(ntr, N, nl - 1), tr_arr_old=(ntr, N, nl - 1), UV=(E, nl - 1, 2), UV_rhsAB=(E, nl - 1, 2),
                eta_n=(N,), d_eta=(N,), ssh_rhs=(N,), ssh_rhs_old=(N,), hbar=(N,), hbar_old=(N,), dhe=(myE,),
                hnode=(N, nl - 1), hnode_new=(N, nl - 1), helem=(myE, nl - 1), zbar_3d_n=(N, nl), Z_3d_n=(N, nl - 1),
                Wvel=(N, nl), Wvel_e=(N, nl), Wvel_i=(N, nl), ssh_values=(d.ssh_nza,))


def _view(ptr, shape):
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return np.zeros(shape, dtype=np.float64 if ptr._type_ is C.c_double else np.int32)
    return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(shape)


class Mesh:
    def __init__(self, handle, lib, opts):
        self._h, self._lib, self.opts = handle, lib, opts
        self.desc_p = lib.fesom_mesh_get_desc(handle)
        self.part_p = lib.fesom_mesh_get_part(handle)
        self.d = self.desc_p.contents
        self.shapes = _shapes(self.d)

    @classmethod
    def load(cls, meshdir, which_ale="zstar", use_partial_cell=True, force_rotation=True, cyclic_length_deg=360.0,
             dt=900.0, alpha=1.0, theta=1.0, K_hor=3000.0, euler=(50.0, 15.0, -90.0), npes=1, mype=0, use_cavity=False, use_cavity_partial_cell=False, cavity_partial_cell_thresh=0.0):
        lib = _lib.load()
        o = _lib.MeshOpts(int(force_rotation), cyclic_length_deg, euler[0], euler[1], euler[2], int(use_partial_cell),
                          WHICH_ALE[which_ale], dt, alpha, theta, K_hor, npes, mype, int(use_cavity), int(use_cavity_partial_cell), cavity_partial_cell_thresh)
        h = lib.fesom_mesh_load(str(meshdir).encode(), C.byref(o))
        if not h:
            raise RuntimeError(f"fesom_mesh_load failed for {meshdir}")
        return cls(h, lib, o)

    def __getattr__(self, name):
        sh = self.__dict__.get("shapes", {})
        if name in sh:
            return _view(getattr(self.d, name), sh[name])
        if name in ("nl", "nod2D", "elem2D", "edge2D", "edge2D_in", "myDim_nod2D", "eDim_nod2D", "myDim_elem2D",
                    "eDim_elem2D", "eXDim_elem2D", "myDim_edge2D", "eDim_edge2D", "ssh_nza", "max_nod_in_elem"):
            return getattr(self.d, name)
        raise AttributeError(name)

    def initial_state(self, num_tracers=2):
        """Zero prognostic state + ALE thickness initialisation (array_setup / init_thickness_ale)."""
        sp = self._lib.fesom_mesh_get_initial_state(self._h, num_tracers)
        s = sp.contents
        shp = state_shapes(self.d, num_tracers)
        return State({k: _view(getattr(s, k), shp[k]).copy() for k in _lib.STATE_FIELDS}, self, num_tracers)

    def level_area_test(self):
        """check_mesh_consistency of the reference (src/oce_mesh.F90:2452-2494): per level, the sum of the nodal control-volume
        areas (areasvol) over the owned wet nodes and the sum of the areas of the wet elements whose first node is owned; the two
        agree to round-off on a consistent mesh.  Returns (vol_n, vol_e), arrays of length nl; rank-local sums on a partition."""
        nl, myN, myE = self.nl, self.myDim_nod2D, self.d.myDim_elem2D
        lev = np.arange(1, nl + 1)[None, :]
        wet_n = (lev >= self.ulevels_nod2D[:myN, None]) & (lev <= self.nlevels_nod2D[:myN, None] - 1)
        vol_n = np.where(wet_n, self.areasvol[:myN], 0.0).sum(axis=0)
        first_owned = self.elem2D_nodes[:myE, 0] <= myN
        wet_e = (lev >= self.ulevels[:myE, None]) & (lev <= self.nlevels[:myE, None] - 1) & first_owned[:, None]
        vol_e = np.where(wet_e, self.elem_area[:myE, None], 0.0).sum(axis=0)
        return vol_n, vol_e

    def total_volume(self, state):
        """check_total_volume (src/oce_mesh.F90:2509-2550): ocean volume from the node columns and from the element columns"""
        nl, myN, myE = self.nl, self.myDim_nod2D, self.d.myDim_elem2D
        lev = np.arange(1, nl)[None, :]
        wet_n = (lev >= self.ulevels_nod2D[:myN, None]) & (lev <= self.nlevels_nod2D[:myN, None] - 1)
        vn = float(np.where(wet_n, self.areasvol[:myN, :nl - 1] * state.hnode[:myN], 0.0).sum())
        first_owned = self.elem2D_nodes[:myE, 0] <= myN
        wet_e = (lev >= self.ulevels[:myE, None]) & (lev <= self.nlevels[:myE, None] - 1) & first_owned[:, None]
        ve = float(np.where(wet_e, self.elem_area[:myE, None] * state.helem[:myE], 0.0).sum())
        return vn, ve

    def wet_counts(self):
        """(N3, E3, D3) = wet node cells, prism cells, edge cells (SURVEY.md conventions)."""
        n3 = int((self.nlevels_nod2D[: self.myDim_nod2D] - 1).sum())
        e3 = int((self.nlevels[: self.myDim_elem2D] - 1).sum())
        et = self.edge_tri[: self.myDim_edge2D]
        l1 = self.nlevels[et[:, 0] - 1] - 1
        l2 = np.where(et[:, 1] > 0, self.nlevels[np.maximum(et[:, 1], 1) - 1] - 1, 0)
        return n3, e3, int(np.maximum(l1, l2).sum())

    def free(self):
        if self._h:
            self._lib.fesom_mesh_free(self._h)
            self._h = None


class State:
    """Prognostic state arrays (numpy, host).  desc() gives the C struct for upload/download."""

    def __init__(self, arrays, mesh, ntr):
        self.a, self.mesh, self.ntr = arrays, mesh, ntr

    def __getattr__(self, k):
        a = self.__dict__.get("a", {})
        if k in a:
            return a[k]
        raise AttributeError(k)

    def desc(self):
        s = _lib.StateDesc()
        for k in _lib.STATE_FIELDS:
            arr = self.a[k]
            assert arr.flags["C_CONTIGUOUS"] and arr.dtype == np.float64
            setattr(s, k, arr.ctypes.data_as(_lib.PD))
        return s
