"""The C-ABI library loads on a GPU-less host and exports exactly what include/ctpvae_radon.h declares."""
import os
import re
import subprocess

import pytest

from tests.conftest import ROOT

HEADER = os.path.join(ROOT, "include", "ctpvae_radon.h")


@pytest.fixture(scope="module")
def built_lib():
    from ct_pvae_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "ct_pvae_amd", "csrc"), "-s"], check=True)
    return _lib


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(ctpvae_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    names = declared_symbols()
    for must in ("ctpvae_rotate_fwd_f32", "ctpvae_rotate_bwd_f32", "ctpvae_rotate_transforms_f32",
                 "ctpvae_siddon_fwd_f32", "ctpvae_fbp_filter_f64", "ctpvae_fbp_backproject_f64",
                 "ctpvae_loglik_fwd_f32", "ctpvae_loglik_bwd_f32", "ctpvae_num_proj_pix", "ctpvae_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(built_lib):
    out = subprocess.run(["nm", "-D", "--defined-only", built_lib.LIB_PATH], check=True, capture_output=True,
                         text=True).stdout
    exported = set(re.findall(r"\bT (ctpvae_[a-z0-9_]+)", out))
    assert exported == set(declared_symbols())


def test_binding_table_matches_header(built_lib):
    assert sorted(built_lib.SIGNATURES) == declared_symbols()
    lib = built_lib.load()
    macro = int(re.search(r"#define\s+CTPVAE_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert lib.ctpvae_abi_version() == built_lib.ABI_VERSION == macro == 3400


def test_torch_node_refuses_a_library_of_another_abi(built_lib, monkeypatch):
    """the C++ autograd node resolves entry points by name, so it must compare the ABI it was compiled for with the
    library it binds -- a stale node falls back to the Python nodes with a warning instead of calling with old arguments."""
    import warnings
    if not os.path.exists(built_lib.NODE_PATH):
        pytest.skip("torch node not built")
    lib = built_lib.load()

    class Other:
        def __getattr__(self, name):
            return getattr(lib, name)

        def ctpvae_abi_version(self):
            return 2001
    monkeypatch.setattr(built_lib, "_node", False)
    monkeypatch.setattr(built_lib, "_lib", Other())
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert built_lib.torch_node() is None
    assert any("built for ABI" in str(x.message) for x in w)
    monkeypatch.setattr(built_lib, "_node", False)


def test_host_only_entry_points(built_lib):
    import ctypes
    lib = built_lib.load()
    assert lib.ctpvae_num_proj_pix(128, 128) == 184
    assert lib.ctpvae_num_proj_pix(512, 512) == 728
    lo, hi = ctypes.c_int(), ctypes.c_int()
    assert lib.ctpvae_pad_amounts(128, 184, ctypes.byref(lo), ctypes.byref(hi)) == 0
    assert (lo.value, hi.value) == (28, 28)
    assert lib.ctpvae_siddon_dx(128, 128, 1) == 184 and lib.ctpvae_siddon_dx(128, 128, 0) == 128


def test_tiled_workspace_rule(built_lib):
    """Host-only size rule of the tiled forward: 0 when the slice fits LDS whole, else S x tiles x A x slots fp32 (512x512
    nearest: 8 x 4 tiles of 64 x 128 -- equal rows of tiles, ABI 3310 --, 192 slots; bilinear, ABI 3400: 8 x 6 tiles of 64 x 86
    with a one-pixel halo, 128 slots)."""
    lib = built_lib.load()
    assert built_lib.tile_shape(512, 512) == (128, 64) and built_lib.tile_shape(1000, 1024) == (125, 64)
    assert built_lib.tile_shape(300, 200) == (100, 64) and built_lib.tile_shape(96 * 3, 640) == (96, 64)
    assert built_lib.tile_shape(128, 128) is None and built_lib.tile_shape(128, 128, 1) is None
    assert built_lib.tile_shape(512, 512, 1) == (86, 64) and built_lib.tile_shape(300, 200, 1) == (75, 64)
    assert built_lib.tile_shape(184, 184, 1) is None and built_lib.tile_shape(192, 192, 1) == (96, 64)
    # the checker restates the rule (oracle.tile_shape) instead of reading it from the library under test: the two agree
    from oracle import radon_oracle
    for H, W in ((512, 512), (1000, 1024), (300, 200), (288, 640), (700, 100), (2000, 64)):     # geometries that ARE tiled
        for interp in (0, 1):
            assert built_lib.tile_shape(H, W, interp) == radon_oracle.tile_shape(H, W, interp), (H, W, interp)
    with built_lib.tuned("TILED_TH", 96):     # a timing-build switch since round 5: the product library ignores it (it would change bits)
        assert built_lib.tile_shape(512, 512) == (128, 64)
    with pytest.raises(ValueError):
        built_lib.tile_shape(0, 512)
    assert lib.ctpvae_rotate_fwd_tiled_workspace_bytes(50, 128, 128, 184, 184, 20, 0) == 0
    assert lib.ctpvae_rotate_fwd_tiled_workspace_bytes(8, 512, 512, 728, 728, 90, 1) == 8 * 48 * 90 * 128 * 4
    assert lib.ctpvae_rotate_fwd_tiled_workspace_bytes(50, 128, 128, 184, 184, 20, 1) == 0
    assert lib.ctpvae_rotate_fwd_tiled_workspace_bytes(8, 512, 512, 728, 728, 90, 0) == 8 * 32 * 90 * 192 * 4
    assert lib.ctpvae_rotate_fwd_tiled_workspace_bytes(0, 512, 512, 728, 728, 90, 0) == built_lib.EINVAL


def test_bad_arguments_are_reported_not_thrown(built_lib):
    lib = built_lib.load()
    assert lib.ctpvae_num_proj_pix(0, 5) == built_lib.EINVAL
    assert "positive" in built_lib.last_error()
    # null pointers are rejected before any HIP call, so this is safe without a GPU
    rc = lib.ctpvae_rotate_fwd_f32(None, 1, 8, 8, 8, 8, 0, 0, None, 1, 0, None, None)
    assert rc == built_lib.EINVAL and "null" in built_lib.last_error()
    with pytest.raises(ValueError):
        built_lib.check(rc, "rotate_fwd")


def test_siddon_tables_match_oracle(built_lib, oracle):
    """Host-side table builder of the product (libtomo's fmodf/sinf/cosf/quadrant) vs the oracle's inline one:
    the oracle's toy and axis-aligned answers depend on these."""
    import numpy as np
    lib = built_lib.load()
    theta = np.concatenate([np.linspace(0, np.pi, 180, endpoint=False), [-0.3, 3.5, 7.0, 2 * np.pi]]).astype(np.float32)
    s, c = np.empty_like(theta), np.empty_like(theta)
    q = np.empty(theta.size, np.int32)
    assert lib.ctpvae_siddon_tables_f32(theta.ctypes.data, theta.size, s.ctypes.data, c.ctypes.data, q.ctypes.data) == 0
    tp = np.fmod(theta, np.float32(2 * np.pi))
    np.testing.assert_allclose(s, np.sin(tp.astype(np.float64)), atol=1e-7)
    np.testing.assert_allclose(c, np.cos(tp.astype(np.float64)), atol=1e-7)
    want_q = (((tp >= 0) & (tp < np.pi / 2)) | ((tp >= np.pi) & (tp < 1.5 * np.pi)) |
              ((tp < 0) & (((tp + 2 * np.pi) < np.pi / 2) | (((tp + 2 * np.pi) >= np.pi) & ((tp + 2 * np.pi) < 1.5 * np.pi)))))
    away = np.abs(np.mod(tp, np.pi / 2)) > 1e-5
    np.testing.assert_array_equal(q[away], want_q[away].astype(np.int32))


def test_driver_build_entry_point():
    """__graft_entry__.build() is what the driver runs on the GPU-less host every round: it must compile, load the
    library and agree with it about the ABI version."""
    import __graft_entry__ as entry
    entry.build()


def test_round2_host_only_size_rules(built_lib):
    """Host-side size rules of the round-2 entry points (no GPU needed): the exact-transpose plan is the backward plan's
    layout over 2A virtual angles plus one overflow line; the back-projector's workspace holds one partial image per angle
    group; bad sizes are refused."""
    lib = built_lib.load()
    # 128 x 128 in a 184 x 184 canvas, 20 angles: ceil(40 / 16) = 3 groups x 128 rows x 128 padded columns x 16 B + 256
    assert lib.ctpvae_rotate_exact_plan_bytes(128, 128, 184, 184, 20) == 3 * 128 * 128 * 16 + 256
    assert lib.ctpvae_rotate_exact_plan_bytes(128, 128, 184, 184, 180) == 23 * 128 * 128 * 16 + 256
    assert lib.ctpvae_rotate_exact_plan_bytes(512, 512, 728, 728, 90) == 0          # bins do not fit a byte
    assert lib.ctpvae_rotate_exact_plan_bytes(0, 128, 184, 184, 20) == built_lib.EINVAL
    # the ray table (16 B per ray), one flag per angle, one bit per (pixel, angle), one scratch image per slice -- 256-B aligned
    assert lib.ctpvae_siddon_bwd_workspace_bytes(50, 184, 184, 180, 184) == 180 * 184 * 16 + 768 + 6 * 184 * 184 * 4 + 50 * 184 * 184 * 4
    assert lib.ctpvae_siddon_bwd_workspace_bytes(0, 64, 64, 16, 94) == built_lib.EINVAL


def test_no_kernel_spills_registers():
    """Every gfx950 kernel of the library fits its register budget: hipcc's own resource report (-Rpass-analysis) shows no
    scratch for any of them.  A spilled four-slice tile kernel ran 185 us instead of 107 (round 4) -- and still passed every
    parity test."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    csrc = os.path.join(ROOT, "ct_pvae_amd", "csrc")
    bad = []
    for src in ("rotate_cplan.hip", "rotate_plan.hip", "rotate.hip", "rotate_bilin.hip"):
        out = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                              "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull],
                             cwd=csrc, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        name = None
        for line in out.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and int(m.group(1)) > 0:
                bad.append((src, name, int(m.group(1))))
    assert not bad, f"kernels with scratch (register spills): {bad}"


def test_no_device_trap_in_the_projector_kernels():
    """Round 3 guarded "the dynamic LDS starts at address 0" with __builtin_trap() inside the planned / compact kernels: had it
    ever fired, the caller's process would have died with a GPU abort.  The guard is a host-side check now
    (CTPVAE_REQUIRE_NO_STATIC_LDS) and the kernels hold no s_trap."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    csrc = os.path.join(ROOT, "ct_pvae_amd", "csrc")
    for src, kernels in (("rotate_cplan.hip", ("rotate_fwd_compact_kernel",)),
                         ("rotate_plan.hip", ("rotate_fwd_planned_kernel", "rotate_bwd_planned_kernel", "rotate_bwd_planned_sel_kernel"))):
        out = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                              "--cuda-device-only", "-S", src, "-o", "-"], cwd=csrc, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        bodies = re.findall(r"^(_ZN6ctpvae\w+):.*?s_endpgm", out.stdout, re.S | re.M)
        assert bodies
        for m in re.finditer(r"^(_ZN6ctpvae\w+):(.*?)s_endpgm", out.stdout, re.S | re.M):
            if any(k in m.group(1) for k in kernels):
                assert "s_trap" not in m.group(2), m.group(1)
