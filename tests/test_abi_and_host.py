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
