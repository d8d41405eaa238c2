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
    src = '#include <stdio.h>\n#include "fesom_gpu.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(fesom_mesh_desc), sizeof(fesom_com_desc), sizeof(fesom_part_desc), sizeof(fesom_params), sizeof(fesom_state_desc), sizeof(fesom_forcing_desc), sizeof(fesom_mesh_opts));return 0;}'
    exe = "/tmp/_fesom_sizes"
    subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), "-x", "c", "-", "-o", exe], input=src, text=True, check=True)
    sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    mine = [C.sizeof(t) for t in (_lib.MeshDesc, _lib.ComDesc, _lib.PartDesc, _lib.Params, _lib.StateDesc, _lib.ForcingDesc, _lib.MeshOpts)]
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
