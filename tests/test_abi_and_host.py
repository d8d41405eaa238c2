"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/fesom_gpu.h declares, and refuses to
run without a GPU (no CPU fallback in the product path)."""
import ctypes as C
import os
import re
import subprocess
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PI = os.path.join(REPO, "tests", "golden", "meshes", "pi")


def _declared():
    txt = open(os.path.join(REPO, "include", "fesom_gpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"^\s*(?:int|void|double|const\s+\w+\s*\*|void\s*\*|const char\s*\*)\s*\*?\s*(\w+)\s*\(", txt, flags=re.M)
    return sorted(set(n for n in names if n.startswith(("fesom_", "psolve"))))


def test_exports_match_header(built):
    from fesom2_amd import _lib
    decl = _declared()
    assert len(decl) >= 20
    assert sorted(_lib.EXPORTS) == decl, (sorted(set(decl) ^ set(_lib.EXPORTS)))
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert set(decl) <= exported, set(decl) - exported
    lib = _lib.load()
    for s in decl:
        assert hasattr(lib, s)


def test_gfx950_code_object(built):
    from fesom2_amd import _lib
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data and b"k_tr_update" in data


def test_struct_layout_matches_header(built):
    """ctypes mirrors must have the size the C compiler gives the header structs"""
    from fesom2_amd import _lib
    src = '#include <stdio.h>\n#include "fesom_gpu.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(fesom_mesh_desc), sizeof(fesom_com_desc), sizeof(fesom_part_desc), sizeof(fesom_params), sizeof(fesom_state_desc), sizeof(fesom_forcing_desc), sizeof(fesom_mesh_opts), sizeof(fesom_step_info), sizeof(fesom_transport));return 0;}'
    exe = "/tmp/_fesom_sizes"
    subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), "-x", "c", "-", "-o", exe], input=src, text=True, check=True)
    sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    mine = [C.sizeof(t) for t in (_lib.MeshDesc, _lib.ComDesc, _lib.PartDesc, _lib.Params, _lib.StateDesc, _lib.ForcingDesc, _lib.MeshOpts, _lib.StepInfo, _lib.Transport)]
    assert sizes == mine, (sizes, mine)


def test_no_gpu_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    mesh = Mesh.load(PI, dt=900.0)
    try:
        OceanCore(mesh, make_params())
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("fesom_gpu_init must fail without a HIP device")
    lib = mesh._lib
    assert lib.fesom_gpu_step(1) != 0 and lib.fesom_gpu_call(b"pressure_bv", 0) != 0


def test_product_never_touches_oracle():
    """nothing under fesom2_amd/ may import, link or open anything under oracle/"""
    for root, _, files in os.walk(os.path.join(REPO, "fesom2_amd")):
        if "build" in root.split(os.sep) or "__pycache__" in root:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "liborc" not in txt and "orc_" not in txt and "oracle_lib" not in txt, f
                assert not re.search(r"(import|from)\s+oracle", txt), f


def test_mesh_counts_and_wet_cells(built):
    from fesom2_amd.mesh import Mesh
    mesh = Mesh.load(PI, dt=900.0)
    assert (mesh.nod2D, mesh.elem2D, mesh.edge2D, mesh.nl) == (3140, 5839, 8986, 48)
    assert mesh.wet_counts() == (101749, 182245, 284123)          # SURVEY.md conventions: N3, E3, D3 of the pi mesh
    s = Mesh.load(os.path.join(REPO, "tests", "golden", "meshes", "soufflet"), force_rotation=False, cyclic_length_deg=4.5, dt=1200.0)
    assert (s.nod2D, s.elem2D, s.edge2D, s.nl) == (2875, 5700, 8575, 41)
    assert s.wet_counts() == (115000, 228000, 343000)


def test_generated_edges_equal_mesh_files(built, monkeypatch):
    """edge generation of the host mesh layer (for generated meshes that ship without edge files) = the reference
    partitioner's find_edges_ini (src/fvom_init.F90:315-650): on the reference's own meshes it must reproduce their
    edges.out / edge_tri.out / edgenum.out entry for entry"""
    from fesom2_amd.mesh import Mesh
    for name, kw in (("pi", dict(dt=900.0)), ("soufflet", dict(force_rotation=False, cyclic_length_deg=4.5, dt=1200.0))):
        d = os.path.join(REPO, "tests", "golden", "meshes", name)
        a = Mesh.load(d, **kw)
        ea, ta, da = a.edges.copy(), a.edge_tri.copy(), (a.edge2D, a.edge2D_in)
        monkeypatch.setenv("FESOM_MESH_GENERATE_EDGES", "1")
        b = Mesh.load(d, **kw)
        monkeypatch.delenv("FESOM_MESH_GENERATE_EDGES")
        assert (b.edge2D, b.edge2D_in) == da and (ea == b.edges).all() and (ta == b.edge_tri).all(), name
        a.free(); b.free()


def test_refined_mesh_is_consistent(built, tmp_path):
    """uniform refinement (fesom2_amd/mesh_refine.py): 4x the elements, Euler characteristic and total area preserved"""
    import numpy as np
    from fesom2_amd import mesh_refine
    from fesom2_amd.mesh import Mesh
    pi = os.path.join(REPO, "tests", "golden", "meshes", "pi")
    a = Mesh.load(pi, dt=900.0)
    N, E = mesh_refine.refine(pi, str(tmp_path / "r1"), 1)
    b = Mesh.load(str(tmp_path / "r1"), dt=900.0)
    assert E == 4 * a.elem2D and b.elem2D == E and b.nod2D == N == a.nod2D + a.edge2D
    assert b.edge2D == 2 * a.edge2D + 3 * a.elem2D
    assert abs(b.elem_area.sum() / a.elem_area.sum() - 1.0) < 2e-3          # (flat triangles on the sphere: not exactly additive)
    assert (b.nlevels_nod2D >= 2).all() and b.nl == a.nl
    a.free(); b.free()


def test_level_area_and_volume_checks_match_reference_printout(built):
    """check_mesh_consistency / check_total_volume (src/oce_mesh.F90:2452-2550) restated on the host mesh layer, against the numbers
    the reference itself printed on the pi mesh (tests/golden/level_area_test_pi.json)"""
    import json
    from fesom2_amd.mesh import Mesh
    mesh = Mesh.load(PI, dt=900.0)
    ref = json.load(open(os.path.join(REPO, "tests", "golden", "level_area_test_pi.json")))
    vn, ve = mesh.level_area_test()
    assert np.allclose(vn, ref["vol_n"], rtol=1e-13, atol=0) and np.allclose(ve, ref["vol_e"], rtol=1e-13, atol=0)
    assert np.allclose(vn, ve, rtol=1e-13)
    tn, te = mesh.total_volume(mesh.initial_state(2))
    assert abs(tn - te) < 1e-3 * tn and tn > 1e17        # (node and element columns differ in their partial bottom cells)


def test_explicit_inverse_host_builder_equals_oracle_bitwise():
    """The SSH preconditioner of pi-class operators is built on the host (csrc/precond_host.cpp: row scaling, reverse Cuthill-McKee,
    banded LU, substitution, fp32 rounding, drop rule); the oracle restates it in C (oracle/c/orc_xinv.c).  On the pi operator both
    give the same matrix bit for bit (dense and sparsified), and it is an inverse: max |I - A_s M| < 1e-6 (fp32), 82 entries per row."""
    import sys
    from fesom2_amd import _lib
    from fesom2_amd.mesh import Mesh
    import oracle_lib
    lib = _lib.load()
    oracle_lib.build()
    orc = C.CDLL(oracle_lib.ORC_LIB)
    mesh = Mesh.load(os.path.join(REPO, "tests", "golden", "meshes", "pi"), dt=900.0)
    n = mesh.myDim_nod2D
    rp = (np.array(mesh.ssh_rowptr[:n + 1]) - mesh.ssh_rowptr[0]).astype(np.int32)
    ci = (np.array(mesh.ssh_colind_loc) - 1).astype(np.int32)
    vals = np.ctypeslib.as_array(mesh.desc_p.contents.ssh_values, shape=(int(mesh.ssh_nza),)).copy()
    ld = (n + 255) // 256 * 256
    A, B = np.zeros((n, ld), dtype=np.float32), np.zeros((n, ld), dtype=np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.fesom_xinv_build(n, vp(rp), vp(ci), vp(vals), None, ld, vp(A), None) == 0
    assert orc.orc_xinv_build(n, vp(rp), vp(ci), vp(vals), ld, vp(B)) == 0
    assert np.array_equal(A.view(np.int32), B.view(np.int32))
    sc = 1.0 / np.add.reduceat(np.abs(vals), rp[:-1])
    As = np.zeros((n, n))
    for i in range(n):
        As[i, ci[rp[i]:rp[i + 1]]] = vals[rp[i]:rp[i + 1]] * sc[i]
    assert np.abs(np.eye(n) - As @ A[:, :n].astype(np.float64)).max() < 1e-6
    out = []
    for fn in (lib.fesom_xinv_sparsify, orc.orc_xinv_sparsify):
        fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        mp = np.zeros(n + 1, dtype=np.int32)
        fn(n, ld, vp(A), 1e-4, vp(mp), None, None)
        mc, mv = np.zeros(mp[n], dtype=np.uint16), np.zeros(mp[n], dtype=np.float32)
        fn(n, ld, vp(A), 1e-4, vp(mp), vp(mc), vp(mv))
        out.append((mp, mc, mv))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    assert 60 < out[0][0][n] / n < 110


def test_psolver_init_refuses_a_block_of_a_multi_rank_partition(tmp_path):
    """psolver_init / psolve keep the reference's void signatures (src/psolve.c:16,152), so a violated precondition cannot be
    returned: the library prints one line and exits with status 3 -- a block of a multi-rank partition (row offset, or global
    column indices) and a non-zero rptr[0] are refused before any device work."""
    import sys
    import textwrap
    script = tmp_path / "p.py"
    script.write_text(textwrap.dedent(f"""
        import ctypes as C, sys
        import numpy as np
        sys.path.insert(0, {REPO!r})
        from fesom2_amd import _lib
        lib = _lib.load()
        case = sys.argv[1]
        n = 4
        rptr = np.array([0, 2, 4, 6, 8], dtype=np.int32); cols = np.array([0, 1, 1, 2, 2, 3, 3, 0], dtype=np.int32)
        vals = np.array([2., -1.] * 4); part = np.array([0, n], dtype=np.int32)
        if case == "offset": part = np.array([4, 8], dtype=np.int32)
        if case == "global_cols": cols[1] = 7
        if case == "rptr0": rptr = rptr + 1
        one = C.c_int(1); z = C.c_int(0); tol = C.c_double(1e-10); mi = C.c_int(2000); dt = C.c_double(1e-8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        lib.psolver_init(C.byref(one), C.byref(one), C.byref(one), C.byref(one), C.byref(one), C.byref(one), C.byref(dt), C.byref(mi),
                         C.byref(one), C.byref(tol), p(part), p(rptr), p(cols), p(vals), C.byref(z), C.byref(z))
        print("returned")
    """))
    for case, text in (("offset", "part[0] != 0"), ("global_cols", "column index outside"), ("rptr0", "rptr[0] must be 0")):
        r = subprocess.run([sys.executable, str(script), case], capture_output=True, text=True, timeout=120)
        assert r.returncode == 3 and text in r.stderr and "returned" not in r.stdout, (case, r.returncode, r.stderr[-400:])


def test_psolve_mpi_adapter_compiles_against_the_header():
    """fesom2_amd/fortran/fesom_gpu_psolve_mpi.c (the MPI host adapter of the distributed psolve) against include/fesom_gpu.h and an
    mpi.h: it defines the reference's three entry points (src/psolve.c:16,117,152) and references only symbols the library exports."""
    import shutil
    mpi_inc = next((d for d in ("/opt/conda/include", "/usr/include/mpich", "/usr/include/x86_64-linux-gnu/mpich") if os.path.exists(os.path.join(d, "mpi.h"))), None)
    if mpi_inc is None or shutil.which("gcc") is None:
        pytest.skip("no mpi.h / gcc here")
    src = os.path.join(REPO, "fesom2_amd", "fortran", "fesom_gpu_psolve_mpi.c")
    obj = os.path.join(REPO, "fesom2_amd", "build", "psolve_mpi_check.o")
    os.makedirs(os.path.dirname(obj), exist_ok=True)
    r = subprocess.run(["gcc", "-O1", "-fPIC", "-I", mpi_inc, "-I", os.path.join(REPO, "include"), "-c", src, "-o", obj], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    nm = subprocess.run(["nm", obj], capture_output=True, text=True).stdout
    for sym in ("psolver_init", "psolve", "psolver_final"):
        assert f" T {sym}" in nm, sym
    from fesom2_amd import _lib
    wanted = {ln.split()[-1] for ln in nm.splitlines() if " U fesom_gpu_" in ln}
    assert wanted and wanted <= set(_lib.EXPORTS), wanted - set(_lib.EXPORTS)


def test_psolver_init_dist_validates_before_any_device_work(built):
    """fesom_gpu_psolver_init_dist returns an error with a message (no exit: it has a status) for a rank outside the partition, a row block whose column is
    neither owned nor in the halo list, a send-list entry outside the owned rows -- all checked before a device is touched."""
    import ctypes as C
    from fesom2_amd import _lib
    lib = _lib.load()
    lib.fesom_gpu_last_error.restype = C.c_char_p
    I = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    part = I([0, 4, 8])
    rptr = I([0, 2, 4, 6, 8]); cols = I([0, 1, 1, 2, 2, 3, 3, 4]); vals = np.array([2.0, -1.0] * 4)
    rPE, rcnt, rglob, sPE, scnt, sloc = I([1]), I([1]), I([4]), I([1]), I([1]), I([0])
    lib.fesom_gpu_psolver_init_dist.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_double] + [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] * 2 + [C.c_void_p]

    def call(npes=2, mype=0, cols_=cols, rglob_=rglob, sloc_=sloc):
        return lib.fesom_gpu_psolver_init_dist(npes, mype, P(part), P(rptr), P(cols_), P(vals), 100, 1e-10, 1, P(rPE), P(rcnt), P(rglob_), 1, P(sPE), P(scnt), P(sloc_), None)
    assert call(npes=1) != 0 and b"npes >= 2" in lib.fesom_gpu_last_error()
    assert call(mype=5) != 0 and b"mype" in lib.fesom_gpu_last_error()
    assert call(rglob_=I([5])) != 0 and b"neither owned nor in the halo list" in lib.fesom_gpu_last_error()
    assert call(sloc_=I([9])) != 0 and b"send list entry outside" in lib.fesom_gpu_last_error()
    bad = cols.copy(); bad[0] = 1; bad[1] = 0
    assert call(cols_=bad) != 0 and b"diagonal" in lib.fesom_gpu_last_error()


def test_bench_byte_table_covers_every_kernel_of_the_step():
    """bench.py: every kernel step_kernels() lists for an option set has an entry in KERNEL_VALUES (a missing entry would only show on the GPU box)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)

    class P:
        pass
    for mix, gm, redi, toy, visc in ((1, 1, 1, 0, 5), (1, 1, 0, 0, 5), (1, 0, 1, 0, 5), (2, 0, 0, 0, 5), (2, 0, 0, 1, 5), (2, 0, 0, 0, 7)):
        p = P()
        p.mix_scheme, p.Fer_GM, p.Redi, p.toy_soufflet, p.visc_option = mix, gm, redi, toy, visc
        for tile in (False, True):
            for k in b.step_kernels(p, tile):
                assert k in b.KERNEL_VALUES, k
    for k in b.PER_TRACER + tuple(b.REDI_EXTRA) + tuple(b.SHARED_ONCE) + tuple(b.SHARED_ONCE_REDI):
        assert k in b.KERNEL_VALUES, k


def test_bench_has_no_undefined_names():
    """bench.py only runs end to end on a GPU box: a static pass over its functions for names that are neither bound in the function or an enclosing one
    (arguments, assignments, loop / with / except targets, comprehensions, imports), nor at module level, nor builtins -- the class of slip a CPU-only
    round cannot see otherwise."""
    import ast
    import builtins
    FN = (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)

    def bound(node, top=True):
        """names bound in the scope of `node` itself (nested function bodies excluded, their names included)"""
        names = set()
        if isinstance(node, FN):
            a = node.args
            for x in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                names.add(x.arg)
        stack = list(ast.iter_child_nodes(node))
        while stack:
            n = stack.pop()
            if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                names.add(n.name)
                continue                                   # (another scope)
            if isinstance(n, ast.Lambda):
                continue
            if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
                names.add(n.id)
            elif isinstance(n, (ast.Import, ast.ImportFrom)):
                for al in n.names:
                    names.add((al.asname or al.name).split(".")[0])
            elif isinstance(n, ast.ExceptHandler) and n.name:
                names.add(n.name)
            elif isinstance(n, (ast.Global, ast.Nonlocal)):
                names.update(n.names)
            stack.extend(ast.iter_child_nodes(n))
        return names

    missing = []

    def check(node, outer):
        scope = outer | bound(node)
        stack = list(ast.iter_child_nodes(node))
        while stack:
            n = stack.pop()
            if isinstance(n, FN):
                check(n, scope)
                continue
            if isinstance(n, ast.ClassDef):
                continue
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in scope:
                missing.append(f"{getattr(node, 'name', '<lambda>')}:{n.lineno} {n.id}")
            stack.extend(ast.iter_child_nodes(n))

    for fname in ("bench.py", "__graft_entry__.py", os.path.join("fesom2_amd", "workloads.py"), os.path.join("fesom2_amd", "core.py"), os.path.join("fesom2_amd", "parallel.py")):
        tree = ast.parse(open(os.path.join(REPO, fname)).read())
        top = bound(tree) | set(dir(builtins))
        for n in ast.iter_child_nodes(tree):
            if isinstance(n, FN):
                check(n, top)
            elif isinstance(n, ast.ClassDef):
                for mth in ast.iter_child_nodes(n):
                    if isinstance(mth, FN):
                        check(mth, top)
        assert not missing, (fname, missing)
