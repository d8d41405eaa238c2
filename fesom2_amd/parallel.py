"""Partitioned (multi-GPU) ocean step: one process per GPU, the reference's node partition, halo exchange at the
reference's exchange points (src/gen_halo_exchange.F90; call sites listed below), partitioned SSH solve.

The library (libfesom_gpu.so) packs / unpacks halos and runs the kernels; this host moves the bytes with
`torch.distributed`: backend "nccl" = RCCL over xGMI directly on the device buffers, backend "gloo" = staging through host
memory (used by the 2-rank tests, where both ranks may even share one GPU).  A Fortran/MPI host would drive the same C ABI
phases and call MPI_Isend/Irecv on the device buffers instead.

Sequence = `oce_timestep_ale` (src/oce_ale.F90:2556-2767) for the supported options, with the exchanges the reference
performs where a later phase reads halo values:
  Unode               exchange_nod   oce_dyn.F90:168            Unode_rhs   exchange_nod  oce_ale_vel_rhs.F90:330
  U_b (V_b)           exchange_elem  oce_dyn.F90:613-614        U_c (V_c)   exchange_nod  oce_dyn.F90:636-637
  SSH solve: halo of p, s per SpMV + all-reduce of the dot products (pARMS bicgstab_ras.c)
  d_eta               exchange_nod   oce_ale.F90:2342           UV          exchange_elem oce_dyn.F90:130
  ssh_rhs_old, hbar   exchange_nod   oce_ale.F90:1654,1663      Wvel, hnode_new exchange_nod oce_ale.F90:2135-2136
  tr_xy               exchange_elem (full element halo) oce_tracer_mod.F90:68-72
  fct_LO              exchange_nod   oce_adv_tra_driver.F90:134 fct_plus/minus exchange_nod oce_adv_tra_fct.F90:279
  tr_arr              exchange_nod   oce_ale_tracer.F90:155
(sw_alpha/beta need no exchange here: they are evaluated on halo nodes from the halo T, S; sigma_xy / neutral slope are
leaves without GM/Redi and stay owned-only; Wvel_e, Wvel_i, eta_n, hbar_old travel with Wvel / hbar because the fused node
kernel computes them for owned nodes only.)  Owned values do not depend on the partition except through the SSH solve, whose
dot products are summed in a different order (agreement to the solver tolerance, like the reference across partitions).
"""
import ctypes as C
import os
import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .core import OceanCore
from .mesh import Mesh

NOD, ELEM, ELEM_FULL = 0, 1, 2
MAXITS = 2000


def run_step(core, par, X, solve, first, probe=None, n=1, zonal=None):
    """One step, phase by phase (kernel names of libfesom_gpu.so), in the order of the library's own program of the partitioned step
    (csrc/api.hip:build_program).  X(kind, fields) = halo exchange -- X([(kind, fields), ...]) where node and element fields leave in
    one message --, solve() = SSH solve; `probe(label)` (optional) is called after the phases whose results the tests compare across
    partitions; n = step number and zonal() = the partitioned compute_zonal_mean, both only used by the Soufflet channel hooks.
    The kernels of a step are ordered so that fields that are ready together travel together (13 exchange points with KPP + GM + Redi
    outside the solver, 8 with PP): everything that needs only the incoming state comes first."""
    c, p = core.call, par
    P = probe if probe is not None else (lambda label: None)

    def XM(parts):
        for kind, names in parts:
            X(kind, names)

    c("first_step", 1 if first else 0)
    if p.toy_soufflet and n % 10 == 0:                # before_oce_step (oce_setup_step.F90:625-630)
        zonal() if zonal is not None else c("compute_zonal_mean")
    kpp, v5 = p.mix_scheme == 1, p.visc_option == 5
    ub_early = p.visc_option >= 4                      # (options 1-3 read the Leith coefficient, formed further down)
    c("k_vel_nodes")
    if ub_early:
        c("k_visc_elem"); XM([(NOD, ["Unode"]), (ELEM, ["U_b"])])
    else:
        X(NOD, ["Unode"])
    P("vel_nodes")
    c("k_pressure_bv"); c("k_pgf"); c("k_sigma_slope"); P("pressure")
    if p.use_momix:
        c("k_momix")
    if kpp:
        c("k_kpp_col")
    n2 = []
    if p.mom_adv == 3:
        c("k_vinv_ke"); n2.append("KE_node")
    else:
        c("k_momadv_node"); n2.append("Unode_rhs")
    if v5 and ub_early:
        c("k_visc_node"); n2.append("U_c")
    if p.Redi:
        n2.append("slope_tapered")
    if kpp:
        n2.append("kpp_blmc")
    X(NOD, n2)
    if p.mix_scheme == 2:
        c("k_pp"); P("mixing")
    if kpp:                                           # KPP: smoothing of blmc needs the neighbours' values after every sweep
        c("k_kpp_smooth1"); X(NOD, ["kpp_sA"])
        c("k_kpp_smooth2"); X(NOD, ["kpp_sB"])
        c("k_kpp_smooth3")
        c("k_kpp_final"); X(NOD, ["kpp_viscA", "Kv"])
        c("k_kpp_elem"); P("mixing")
    if p.mom_adv == 3:
        c("k_leith_vort"); X(NOD, ["vorticity"]); c("k_vinv_elem"); P("vel_rhs")
    else:
        c("k_vel_rhs"); P("vel_rhs")
    if p.visc_option <= 3:
        c("k_leith_vort"); X(NOD, ["vorticity"]); c("k_leith_elem")
        for _ in range(2):
            c("k_leith_node"); X(NOD, ["leith_aux"]); c("k_leith_avg")
        X(ELEM, ["Visc"])
        if p.visc_option != 1:
            c("k_visc_elem"); X(ELEM, ["U_b"])
    if not v5:
        c("k_visc_apply")
    c("k_impl_visc"); P("impl_visc")
    if p.which_ale != 0:
        c("k_stiff_update")
    c("k_edge_transport"); c("k_ssh_rhs_node"); P("ssh_rhs")
    solve(); X(NOD, ["d_eta"]); P("solve")
    if p.toy_soufflet:
        c("relax_zonal_vel")                          # oce_ale.F90:2696
    if p.Redi and not p.Fer_GM:
        c("init_Redi_GM"); X(NOD, ["Ki"])
    if p.Fer_GM:                                      # bolus velocities (oce_fer_gm.F90), before vert_vel_ale moves hnode_new
        c("init_Redi_GM"); X(NOD, ["fer_c", "fer_K"] + (["Ki"] if p.Redi else []))
        c("fer_solve_Gamma"); X(NOD, ["fer_gamma"])
        c("fer_gamma2vel"); c("fer_wvel")             # (owned edges only touch elements this rank computes itself: the halos can wait)
    c("k_update_vel"); c("k_edge_transport1"); c("k_vert_vel_hbar")
    nn, ee = ["Wvel", "Wvel_e", "Wvel_i", "hnode_new", "hbar", "hbar_old", "eta_n", "ssh_rhs_old"], ["UV"]
    if p.Fer_GM:
        nn.append("fer_Wvel"); ee.append("fer_UV")
    XM([(NOD, nn), (ELEM, ee)])
    c("k_dhe"); P("vert_vel")
    if p.Fer_GM:
        c("bolus_add")
    if p.SPP:
        c("k_spp", 0)
    c("k_tr_ab", 0); c("k_tr_grad_elem", 0); X(ELEM_FULL, ["tr_xy_ab"])
    c("k_tr_z", 0)
    c("k_updn_grad", 0)
    c("k_flux_hor", 0); c("k_fct_lo_node", 0)
    if not p.tra_adv_lim:                             # (no low-order solution, no limiter with tra_adv_lim='NON')
        X(NOD, ["fct_LO"] + (["tr_z"] if p.Redi else []))
        if p.with_diffusion:
            c("k_diff_flux", 0)
        c("k_fct_node", 0); X(NOD, ["fct_plus", "fct_minus"])
    else:
        if p.Redi:
            X(NOD, ["tr_z"])
        if p.with_diffusion:
            c("k_diff_flux", 0)
    c("k_fct_edge_limit", 0); c("k_tr_update", 0)
    if p.smooth_bh_tra:
        c("k_bh1", 0); X(NOD, ["bh_tmp"]); c("k_bh2", 0)
    if p.toy_soufflet:
        for _ in range(p.num_tracers):                # once per tracer of the loop, always on tracer 1 (oce_ale_tracer.F90:150)
            c("relax_zonal_temp")
    elif p.clim_relax > 1.0e-8:
        c("relax_to_clim", 0)
    X(NOD, ["tr_arr"]); P("tracers")
    if p.Fer_GM:
        c("bolus_remove")
    c("k_thick_node"); c("k_thick_elem"); P("thickness")


class _DevBuf:
    """device buffer exposed to torch through __cuda_array_interface__ (no copy)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class HaloExchanger:
    def __init__(self, core, group=None):
        self.group = group
        self.lib = core.lib
        lib = self.lib
        lib.fesom_gpu_halo_info.argtypes = [C.c_int] + [C.POINTER(C.c_int)] * 8
        lib.fesom_gpu_halo_pack.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        lib.fesom_gpu_halo_unpack.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p)]
        lib.fesom_gpu_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int]
        lib.fesom_gpu_set_stream.argtypes = [C.c_void_p]
        self.device = dist.get_backend() == "nccl"
        self._tens = {}
        self._splits = {}
        if self.device:                       # library kernels and RCCL on the same stream: no host synchronisation anywhere
            self._chk(lib.fesom_gpu_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)), "set_stream")
        self.info = []
        for kind in range(3):
            npes, mype, nr, ns = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            rPE, rc, sPE, sc = ((C.c_int * 4096)() for _ in range(4))       # (one entry per neighbour rank at most)
            rcode = lib.fesom_gpu_halo_info(kind, C.byref(npes), C.byref(mype), C.byref(nr), rPE, rc, C.byref(ns), sPE, sc)
            assert rcode == 0, lib.fesom_gpu_last_error().decode()
            self.info.append(dict(rPE=list(rPE[: nr.value]), rcnt=list(rc[: nr.value]), sPE=list(sPE[: ns.value]), scnt=list(sc[: ns.value])))
        self.npes, self.mype = npes.value, mype.value
        if self.npes > 1:
            self.check_plan()

    def check_plan(self):
        """every message this rank will send is a message its neighbour expects, item for item (the reference's DEBUG check of the
        halo exchange, check_mpi_comm in src/gen_halo_exchange.F90:25-55): an inconsistent partition fails here, not as a hang"""
        mine = [(dict(zip(i["sPE"], i["scnt"])), dict(zip(i["rPE"], i["rcnt"]))) for i in self.info]
        plans = [None] * self.npes
        dist.all_gather_object(plans, mine, group=self.group)
        for kind in range(3):
            for pe, cnt in mine[kind][0].items():
                got = plans[pe][kind][1].get(self.mype)
                if got != cnt:
                    raise RuntimeError(f"halo plan mismatch (kind {kind}): rank {self.mype} sends {cnt} items to {pe}, which expects {got}")
            for pe, cnt in mine[kind][1].items():
                got = plans[pe][kind][0].get(self.mype)
                if got != cnt:
                    raise RuntimeError(f"halo plan mismatch (kind {kind}): rank {self.mype} expects {cnt} items from {pe}, which sends {got}")

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.fesom_gpu_last_error().decode()}")

    def dev_tensor(self, ptr, n):
        key = (ptr, n)
        if key not in self._tens:
            self._tens[key] = torch.as_tensor(_DevBuf(ptr, n), device="cuda")
        return self._tens[key]

    def exchange(self, kind, names):
        if self.npes < 2:
            return
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        sp, rp, W = C.c_void_p(), C.c_void_p(), C.c_int()
        self._chk(self.lib.fesom_gpu_halo_pack(kind, len(names), arr, C.byref(sp), C.byref(rp), C.byref(W)), "halo_pack")
        self.move(kind, sp, rp, W.value)
        self._chk(self.lib.fesom_gpu_halo_unpack(kind, len(names), arr), "halo_unpack")

    def move(self, kind, sp, rp, W):
        """the transport proper: the packed send buffer of the library to the neighbours, their blocks into its receive buffer
        (sp, rp: device addresses as c_void_p; W values per item).  ONE collective per exchange: the packed buffers hold the
        neighbours' blocks in ascending rank order (the order of the reference's com lists), which is the layout of
        all_to_all_single with per-rank split sizes (zero for non-neighbours) -- RCCL turns it into grouped send/recv."""
        inf = self.info[kind]
        ns, nr = sum(inf["scnt"]) * W, sum(inf["rcnt"]) * W
        key = (kind, W)
        if key not in self._splits:
            assert inf["sPE"] == sorted(inf["sPE"]) and inf["rPE"] == sorted(inf["rPE"])
            ss, rs = [0] * self.npes, [0] * self.npes
            for pe, cnt in zip(inf["sPE"], inf["scnt"]):
                ss[pe] = cnt * W
            for pe, cnt in zip(inf["rPE"], inf["rcnt"]):
                rs[pe] = cnt * W
            self._splits[key] = (ss, rs)
        ss, rs = self._splits[key]
        if self.device:                       # RCCL directly on the device buffers
            send, recv = self.dev_tensor(sp.value, max(ns, 1)), self.dev_tensor(rp.value, max(nr, 1))
            dist.all_to_all_single(recv[:nr], send[:ns], rs, ss, group=self.group)
        else:                                 # host staging (gloo)
            send_h, recv_h = np.empty(max(ns, 1)), np.empty(max(nr, 1))
            if ns:
                self._chk(self.lib.fesom_gpu_copy(send_h.ctypes.data, sp, ns * 8, 0), "copy d2h")
            dist.all_to_all_single(torch.from_numpy(recv_h)[:nr], torch.from_numpy(send_h)[:ns], rs, ss, group=self.group)
            if nr:
                self._chk(self.lib.fesom_gpu_copy(rp, recv_h.ctypes.data, nr * 8, 1), "copy h2d")


class PartitionedCore:
    """One rank of a partitioned run.  `torch.distributed` must be initialised (rank = partition index).

    PartitionedCore(meshdir, params, **mesh_kw)  or  PartitionedCore(workload) -- a fesom2_amd.workloads.Workload, whose
    initial state (and forcing) is then uploaded as well.

    transport: "rccl"     the library's built-in transport (ncclSend/ncclRecv groups + ncclAllReduce issued by libfesom_gpu.so
                          on its own stream, include/fesom_gpu.h); default when the process group's backend is nccl;
               "callback" the host moves the bytes (torch.distributed on the device buffers with nccl, host-staged with gloo);
                          default with gloo."""

    def __init__(self, meshdir, params=None, group=None, transport=None, **mesh_kw):
        self.group = group
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        wl = None
        if not isinstance(meshdir, str):
            wl = meshdir
            self.mesh = wl.load_mesh(npes=self.world, mype=self.rank)
            params = wl.params()
        else:
            self.mesh = Mesh.load(meshdir, npes=self.world, mype=self.rank, **mesh_kw)
        self.par = params
        self.core = OceanCore(self.mesh, params)
        self.halo = HaloExchanger(self.core, group)
        self.first = True
        self.solver_iterations = 0
        self._last_its = 8
        self.red_dev = None
        lib = self.core.lib
        lib.fesom_gpu_comm_unique_id.argtypes = [C.c_void_p]
        lib.fesom_gpu_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.fesom_gpu_comm_selftest.argtypes = [C.c_int]
        lib.fesom_gpu_comm_timing.argtypes = [C.c_int]
        lib.fesom_gpu_comm_stats.argtypes = [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]
        self.transport = transport or ("rccl" if dist.get_backend() == "nccl" else "callback")
        if self.transport == "rccl":
            self._init_builtin_transport()
        if self.halo.device:
            lib.fesom_gpu_field_ptr.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_longlong)]
            ptr, cnt = C.c_void_p(), C.c_longlong()
            assert lib.fesom_gpu_field_ptr(b"sv_red", C.byref(ptr), C.byref(cnt)) == 0
            self.red_dev = self.halo.dev_tensor(ptr.value, int(cnt.value))
        self._steps = 0
        self.core.lib.fesom_gpu_toy_zonal_mean.argtypes = [C.c_void_p]
        if wl is not None:
            wl.start(self.core, self.mesh)
            if params.toy_soufflet and self.world > 1:
                self.halo.exchange(ELEM, ["UV"])      # call exchange_elem(UV) of initial_state_soufflet (toy_channel_soufflet.F90:327)
                self.zonal_mean()                     # (wl.start formed rank-local means: redo them over all ranks)

    def _init_builtin_transport(self):
        """RCCL communicator of the library: rank 0 draws the unique id, torch.distributed carries the 128 bytes to the others
        (a Fortran host uses MPI_Bcast), every rank joins with its partition rank; then a self test through the transport."""
        lib = self.core.lib
        idbuf = (C.c_char * 128)()
        if self.rank == 0:
            self.core._chk(lib.fesom_gpu_comm_unique_id(idbuf), "comm_unique_id")
        box = [bytes(idbuf.raw)]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=self.group)
        idbuf.raw = box[0]
        self.core._chk(lib.fesom_gpu_comm_init(idbuf, self.world, self.rank), "comm_init")
        self.core._chk(lib.fesom_gpu_comm_selftest(1000), "comm_selftest")

    @property
    def transport_name(self):
        if self.transport == "rccl":
            lib = os.environ.get("FESOM_GPU_RCCL_LIB")
            what = f"a stand-in for librccl ({os.path.basename(lib)}: test double, host-staged)" if lib else "RCCL"
            return f"built-in: {what} ncclSend/ncclRecv groups + ncclAllReduce issued by the library on its stream"
        return "host callback: torch.distributed " + ("nccl (RCCL) on the device buffers" if self.halo.device else "gloo, host-staged")

    def comm_timing(self, on=True):
        self.core.lib.fesom_gpu_comm_timing(1 if on else 0)

    def comm_stats(self):
        """(exchanges, all-reduces, device ms spent in exchanges [if comm_timing was on]) since the last call"""
        a, b, ms = C.c_longlong(), C.c_longlong(), C.c_double()
        self.core._chk(self.core.lib.fesom_gpu_comm_stats(C.byref(a), C.byref(b), C.byref(ms)), "comm_stats")
        return a.value, b.value, ms.value

    def sync(self):
        self.core.lib.fesom_gpu_sync()

    # -- global sums of the partial dot products (in place in the device buffer sv_red)
    def _allreduce(self, n):
        if self.halo.device:
            dist.all_reduce(self.red_dev[:n], group=self.group)
        else:
            v = self.core.get("sv_red", 8)
            t = torch.from_numpy(v[:n].copy()); dist.all_reduce(t, group=self.group)
            v[:n] = t.numpy(); self.core.set("sv_red", v)

    def solve_ssh(self):
        """Partitioned SSH solve, the phases of csrc/api.hip:part_solve: BiCGstab over the owned rows with the RAS-Chebyshev preconditioner
        of the rank's own block (solver_ras.hip "dsr_", no communication inside it) -- or Jacobi ("ds_") where solver_precond = 0 --, halo of
        the gathered vector before each product with A_s, global sum of the partial dot products after it.  Krylov scalars and the
        convergence flag live on the device; the host reads the flag once per chunk of iterations (phases behind the convergence are
        no-ops, so the result does not depend on the chunks).  A solve that does not converge raises."""
        c, X, AR = self.core.call, self.halo.exchange, self._allreduce
        if self.world == 1:                            # one partition: the single-GPU solver of the library
            c("solve_ssh"); self.solver_iterations = self.core.solver_iterations
            return
        ras = self.core.lib.fesom_gpu_solver_kind() == 2
        c("ds_scale")
        if ras:
            c("dsr_setup"); X(NOD, ["sv_x"])
            c("dsr_init"); AR(1); c("dsr_scal_init")
        else:
            X(NOD, ["sv_dinv"])
            c("ds_setup"); X(NOD, ["sv_s"])
            c("ds_init"); AR(1); c("ds_scal_init"); c("ds_p")
        chunk = max(1, self._last_its + 1)
        while True:
            for _ in range(chunk):
                if ras:
                    c("dsr_prec0"); X(NOD, ["sv_ph"]); c("dsr_spmv1"); AR(1); c("dsr_scal_alpha")
                    c("dsr_prec1"); X(NOD, ["sv_sh"]); c("dsr_spmv2"); AR(4); c("dsr_scal_omega"); c("dsr_update")
                else:
                    X(NOD, ["sv_ph"]); c("ds_spmv1"); AR(1); c("ds_scal_alpha"); c("ds_s")
                    X(NOD, ["sv_s"]); c("ds_spmv2"); AR(4); c("ds_scal_omega"); c("ds_update"); c("ds_p")
            kry = self.core.get("sv_kry", 48)
            if kry[8 if ras else 7] != 0.0 or kry[6] >= MAXITS:
                break
            chunk = 2
        c("dsr_finish" if ras else "ds_finish")
        self.solver_iterations = self._last_its = int(kry[6])
        if not kry[5] < 1e-20:
            raise RuntimeError(f"partitioned SSH solve did not converge: {int(kry[6])} iterations, ||scaled residual|| = {np.sqrt(max(kry[5], 0.0)):.3e}")

    def step(self, n=1, probe=None):
        run_step(self.core, self.par, self.halo.exchange, self.solve_ssh, self.first, probe, n=n, zonal=self.zonal_mean)
        self.first = False

    def zonal_mean(self):
        """compute_zonal_mean of the Soufflet channel over all ranks (rank-local sums on the device, global sums through the
        transport, division): fesom_gpu_toy_zonal_mean"""
        tr = None if (self.transport == "rccl" and self.world > 1) else C.byref(self._get_transport())
        self.core._chk(self.core.lib.fesom_gpu_toy_zonal_mean(tr), "toy_zonal_mean")

    def step_native(self, n=1):
        """The same step driven by the library (fesom_gpu_step_partitioned, include/fesom_gpu.h): the phase loop and the solver
        loop run in C++, this host only supplies the two transport callbacks -- what a Fortran/MPI host does as well
        (fesom2_amd/fortran/fesom_gpu_shim.F90)."""
        self._get_transport()
        self.core.call("first_step", 1 if self.first else 0)
        tr = None if (self.transport == "rccl" and self.world > 1) else C.byref(self._transport)
        self.core._chk(self.core.lib.fesom_gpu_step_partitioned(int(n), tr), "step_partitioned")
        self.first = False
        self.solver_iterations = self.core.lib.fesom_gpu_last_solver_iterations()

    def _get_transport(self):
        if not hasattr(self, "_transport"):
            halo, lib, grp = self.halo, self.core.lib, self.group

            def exchange(ctx, kind, sp, rp, W):
                try:
                    halo.move(kind, C.c_void_p(sp), C.c_void_p(rp), W)
                    return 0
                except Exception as e:          # noqa: BLE001 - must not propagate through the C frame
                    print("transport exchange failed:", e, flush=True)
                    return 1

            def allreduce(ctx, buf, n):
                try:
                    if halo.device:
                        dist.all_reduce(halo.dev_tensor(buf, n), group=grp)
                    else:
                        h = np.empty(n)
                        halo._chk(lib.fesom_gpu_copy(h.ctypes.data, C.c_void_p(buf), n * 8, 0), "copy d2h")
                        t = torch.from_numpy(h); dist.all_reduce(t, group=grp)
                        halo._chk(lib.fesom_gpu_copy(C.c_void_p(buf), h.ctypes.data, n * 8, 1), "copy h2d")
                    return 0
                except Exception as e:          # noqa: BLE001
                    print("transport allreduce failed:", e, flush=True)
                    return 1

            self._cb = (_lib.TRANSPORT_EXCHANGE(exchange), _lib.TRANSPORT_ALLREDUCE(allreduce))     # keep the thunks alive
            self._transport = _lib.Transport(None, self._cb[0], self._cb[1])
        return self._transport

    def owned(self, name, width):
        """(global ids, values) of the owned part of a node field with `width` values per node"""
        n = self.mesh.myDim_nod2D
        N = n + self.mesh.eDim_nod2D
        a = self.core.get(name, N * width).reshape(N, width)
        return self.mesh.myList_nod2D[:n].copy(), a[:n].copy()

    def close(self):
        if self.transport == "rccl":
            self.core.lib.fesom_gpu_comm_finalize()
        self.core.close()
