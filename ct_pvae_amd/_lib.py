"""ctypes binding of the C ABI in include/ctpvae_radon.h (libctpvae_radon.so, built by csrc/Makefile).

This is the only door between the Python host code and the HIP kernels.  There is no CPU fallback: if the
shared library is missing, or a call fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libctpvae_radon.so")

NEAREST, BILINEAR = 0, 1
BWD_TF_COMPAT, BWD_EXACT = 0, 1
EINVAL, EHIP, ENODEV = -1, -2, -3
ABI_VERSION = 3400   # ctpvae_abi_version() of the library this binding was written for

_c_int, _c_float, _vp = ctypes.c_int, ctypes.c_float, ctypes.c_void_p
_ip = ctypes.POINTER(ctypes.c_int)

# name -> (restype, argtypes).  Every symbol the header declares must appear here (tests/test_abi.py
# checks this table against the header and against the built library).
SIGNATURES = {
    "ctpvae_abi_version": (_c_int, []),
    "ctpvae_last_error": (ctypes.c_char_p, []),
    "ctpvae_device_count": (_c_int, []),
    "ctpvae_tune_set": (_c_int, [ctypes.c_char_p, _c_int]),
    "ctpvae_tune_active": (_c_int, []),
    "ctpvae_num_proj_pix": (_c_int, [_c_int, _c_int]),
    "ctpvae_pad_amounts": (_c_int, [_c_int, _c_int, _ip, _ip]),
    "ctpvae_rotate_transforms_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _vp, _vp]),
    "ctpvae_rotate_transforms_host_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _vp]),
    "ctpvae_rotate_fwd_planned_sel_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _vp, _vp]),
    "ctpvae_rotate_fwd_planned_loglik_sel_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int,
                                                          _vp, _vp, _c_int, _vp, _c_float, _vp, _vp, _vp, _vp]),
    "ctpvae_rotate_cplan_supported": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int]),
    "ctpvae_rotate_cplan_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "ctpvae_rotate_cplan_build_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    "ctpvae_rotate_cplan_overflowed": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ctpvae_rotate_fwd_compact_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _vp,
                                               _vp, _c_int, _vp, _c_float, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ctpvae_loglik_tasks_per_row": (_c_int, [_c_int, _c_int]),
    "ctpvae_loglik_part_floats": (ctypes.c_longlong, [_c_int, _c_int, _c_int, _c_int]),
    "ctpvae_loglik_object_sums_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    "ctpvae_rotate_bwd_sel_scaled_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _c_int, _c_int,
                                                  _c_int, _vp, ctypes.c_longlong, _vp, _vp]),
    "ctpvae_rotate_bwd4_plan_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "ctpvae_rotate_bwd4_plan_build_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    "ctpvae_rotate_bwd_planned_sel_scaled_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int,
                                                          _c_int, _vp, ctypes.c_longlong, _vp, _vp]),
    "ctpvae_rotate_fwd_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int,
                                       _c_int, _vp, _vp]),
    "ctpvae_rotate_fwd_f64": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int,
                                       _c_int, _vp, _vp]),
    "ctpvae_rotate_tile_shape": (_c_int, [_c_int, _c_int, _c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "ctpvae_rotate_fwd_tiled_workspace_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int]),
    "ctpvae_rotate_fwd_tiled_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _vp,
                                             _vp, _vp]),
    "ctpvae_rotate_fwd_tiled_interp_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int,
                                                    _vp, _vp, _vp]),
    "ctpvae_rotate_fwd_tiled_loglik_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int,
                                                    _vp, _vp, _vp, _vp, _c_float, _vp, _vp, _vp, _vp]),
    "ctpvae_rotate_tplan_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "ctpvae_rotate_tplan_build_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    "ctpvae_rotate_tplan_overflowed": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ctpvae_rotate_fwd_tiled_compact_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _vp,
                                                     _vp, _vp, _vp, _vp, _c_float, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ctpvae_rotate_bwd_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _c_int,
                                       _c_int, _c_int, _vp, _vp]),
    "ctpvae_rotate_bwd_scaled_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _c_int,
                                              _c_int, _c_int, _vp, ctypes.c_longlong, _vp, _vp]),
    "ctpvae_rotate_plan_supported": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int]),
    "ctpvae_rotate_plan_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int]),
    "ctpvae_rotate_plan_build_f32": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp,
                                              _vp]),
    "ctpvae_rotate_fwd_planned_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp]),
    "ctpvae_rotate_fwd_planned_loglik_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp,
                                                      _c_float, _vp, _vp, _vp, _vp]),
    "ctpvae_rotate_bwd_planned_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp]),
    "ctpvae_rotate_bwd_planned_scaled_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp,
                                                      ctypes.c_longlong, _vp, _vp]),
    "ctpvae_rotate_exact_plan_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "ctpvae_rotate_exact_plan_build_f32": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    "ctpvae_rotate_exact_plan_overflowed": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ctpvae_rotate_bwd_exact_planned_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp]),
    "ctpvae_rotate_exact_wplan_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int]),
    "ctpvae_rotate_exact_wplan_build_f32": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    "ctpvae_rotate_exact_wplan_overflowed": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp]),
    "ctpvae_rotate_bwd_exact_wplan_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _vp,
                                                              _vp, _vp]),
    "ctpvae_rotate_bwd_step_plan_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int]),
    "ctpvae_rotate_bwd_step_plan_build_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    "ctpvae_rotate_bwd_step_plan_overflowed": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp]),
    "ctpvae_rotate_bwd_stepped_scaled_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp,
                                                      ctypes.c_longlong, _vp, _vp]),
    "ctpvae_siddon_dx": (_c_int, [_c_int, _c_int, _c_int]),
    "ctpvae_siddon_tables_f32": (_c_int, [_vp, _c_int, _vp, _vp, _vp]),
    "ctpvae_siddon_fwd_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, _c_int, _c_int, _c_float, _vp,
                                       _vp]),
    "ctpvae_siddon_bwd_workspace_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "ctpvae_siddon_bwd_prepare_f32": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, _c_int, _c_int, _c_float, _vp, _vp]),
    "ctpvae_siddon_bwd_prepared_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, _c_int, _c_int, _c_float, _vp, _vp, _vp,
                                                _vp]),
    "ctpvae_siddon_fwd_workspace_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int]),
    "ctpvae_siddon_fwd_ws_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, _c_int, _c_int, _c_float, _vp, _vp, _vp, _vp,
                                          _vp]),
    "ctpvae_siddon_fwd_ws_tv_dual_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, _c_int, _c_int, _c_float, _vp, _vp, _vp,
                                                  _vp, _vp]),
    "ctpvae_siddon_bwd_tv_primal_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, _c_int, _c_int, _c_float, _vp, _vp,
                                                 _c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ctpvae_siddon_fwd_resid_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, _c_int, _c_int, _c_float, _vp, _vp, _vp,
                                             _vp]),
    "ctpvae_siddon_bwd_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, _c_int, _c_int, _c_float, _vp, _vp, _vp]),
    "ctpvae_siddon_rownorm_f32": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, _c_int, _c_int, _c_float, _vp, _vp]),
    "ctpvae_fbp_filter_f64": (_c_int, [_vp, _c_int, _c_int, _vp, _vp, _vp]),
    "ctpvae_fbp_backproject_f64": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _vp, _vp]),
    "ctpvae_fbp_backproject_bwd_f64": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int, ctypes.c_double,
                                                ctypes.c_double, ctypes.c_double, _vp, _vp]),
    "ctpvae_fbp_backproject_geom_f64": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int, ctypes.c_double,
                                                 ctypes.c_double, ctypes.c_double, _vp, _vp]),
    "ctpvae_gridrec_tables_bytes": (ctypes.c_longlong, [_c_int, _c_int]),
    "ctpvae_gridrec_tables_host_f32": (_c_int, [_c_int, _c_int, _c_float, _vp, _c_int, _vp, _vp]),
    "ctpvae_gridrec_workspace_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int]),
    "ctpvae_gridrec_f32": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp, _vp]),
    "ctpvae_loglik_fwd_f32": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, _vp, _c_float, _vp, _vp]),
    "ctpvae_poisson_measure_f32": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_float, ctypes.c_ulonglong, _vp, _vp]),
    "ctpvae_philox4x32_10": (_c_int, [_vp, _vp, _vp]),
    "ctpvae_loglik_bwd_f32": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _vp, _c_float, _vp, _vp, _vp]),
}

_lib = None


class RadonLibraryError(RuntimeError):
    """The HIP extension is missing or a HIP call inside it failed."""


def load():
    """Load libctpvae_radon.so (once).  Raises RadonLibraryError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RadonLibraryError(
                f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library is stale
            fn.restype = res
            fn.argtypes = args
        if lib.ctpvae_abi_version() != ABI_VERSION:
            raise RadonLibraryError(f"{LIB_PATH} has ABI {lib.ctpvae_abi_version()}, this binding needs {ABI_VERSION}: rebuild it")
        _lib = lib
    return _lib


NODE_NAME = "_ctpvae_torch_node"
NODE_DIR = os.path.join(_HERE, "_torch_node")
NODE_PATH = os.path.join(NODE_DIR, NODE_NAME + ".so")
_node = False


def _node_stamp(src, hdr):
    """What a built node is good for: the ABI of this binding, a hash of its source and of the header it binds, torch's version."""
    import hashlib
    import torch
    h = hashlib.sha256()
    for path in (src, hdr):
        with open(path, "rb") as f:
            h.update(f.read())
    return f"abi={ABI_VERSION} src={h.hexdigest()[:32]} torch={torch.__version__}\n"


def build_torch_node(verbose=False):
    """Compile csrc/torch_node.cpp (host C++, no HIP) with torch's extension builder and put the module in-tree;
    __graft_entry__.build() calls this.  Returns nothing: torch_node() does the one import of the finished file.

    Staleness is decided WITHOUT importing the existing .so (a stale module imported for the probe stays cached by
    (name, path) in CPython and by path in dlopen, so the rebuilt file could never be loaded by this process): a sidecar
    stamp next to the .so holds the ABI, a hash of the source + header and torch's version it was built for.

    The compile runs in a build directory of THIS process (a temporary sibling of NODE_DIR) and the finished .so is moved
    into NODE_DIR with os.replace(): no lock file is ever shared, so a build that was killed cannot leave a baton that a
    later build waits on forever (torch's FileBaton.wait() has no timeout), and two processes building at once each link
    their own file and the last rename wins -- both are the same module."""
    import shutil
    import subprocess
    import sys
    import tempfile
    os.makedirs(NODE_DIR, exist_ok=True)
    src = os.path.join(_HERE, "csrc", "torch_node.cpp")
    hdr = os.path.join(os.path.dirname(_HERE), "include", "ctpvae_radon.h")
    stamp_path = NODE_PATH + ".stamp"
    stamp = _node_stamp(src, hdr)
    if os.path.exists(NODE_PATH) and os.path.exists(stamp_path):
        with open(stamp_path) as f:
            if f.read() == stamp:
                return                           # up to date: nothing to compile, nothing imported
    tmp = tempfile.mkdtemp(prefix=".build-", dir=os.path.dirname(NODE_DIR))
    try:
        # the compile (and the load torch's builder insists on) happen in a CHILD: this process never maps the temporary file
        code = ("import sys; from torch.utils.cpp_extension import load; "
                f"load(name={NODE_NAME!r}, sources=[{src!r}], build_directory={tmp!r}, extra_include_paths=[{os.path.dirname(hdr)!r}], "
                f"extra_cflags=['-O2'], extra_ldflags=['-ldl'], with_cuda=False, verbose={bool(verbose)!r}, is_python_module=False)")
        subprocess.run([sys.executable, "-c", code], check=True)
        os.replace(os.path.join(tmp, NODE_NAME + ".so"), NODE_PATH)
        with open(stamp_path, "w") as f:
            f.write(stamp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for stale in ("lock", "build.ninja", ".ninja_log", ".ninja_deps"):   # leftovers of the round-2/3 in-place builds
        try:
            os.remove(os.path.join(NODE_DIR, stale))
        except OSError:
            pass


def torch_node():
    """The C++ autograd node of the training layout (csrc/torch_node.cpp), bound to the loaded library; None when it was not
    built (the Python node of forward_functions.py then makes the same two C-ABI calls, a few microseconds slower)."""
    global _node
    if _node is False:
        _node = None
        if os.path.exists(NODE_PATH):
            import importlib.util
            import torch  # noqa: F401  (the module links against libtorch)
            try:
                spec = importlib.util.spec_from_file_location(NODE_NAME, NODE_PATH)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
            except (ImportError, OSError) as e:      # built against another libtorch: say so, keep the Python nodes
                import warnings
                warnings.warn(f"{NODE_PATH} does not load ({e}); rebuild it with __graft_entry__.build()")
                return None
            lib = load()
            # the node resolves entry points by NAME: one compiled against another version of the header would call them
            # with that version's argument list.  Compare what it was compiled for with the library that is loaded.
            built_for = mod.compiled_abi() if hasattr(mod, "compiled_abi") else None
            if built_for != lib.ctpvae_abi_version():
                import warnings
                warnings.warn(f"{NODE_PATH} was built for ABI {built_for}, {LIB_PATH} has ABI {lib.ctpvae_abi_version()}: "
                              "using the Python autograd nodes; rebuild it with __graft_entry__.build()")
                return None
            mod.bind(LIB_PATH)
            _node = mod
    return _node


def tune(name, value=-1):
    """Developer knob of the library (see ctpvae_tune_set in include/ctpvae_radon.h); value < 0 unsets, name "*" unsets all."""
    check(load().ctpvae_tune_set(name.encode(), int(value)), "tune_set")


class tuned:
    """`with tuned("NO_PLAN", 1): ...` -- a developer knob set for a block and unset in a finally clause, whatever the block raises."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        tune(self.name, self.value)
        return self

    def __exit__(self, *exc):
        tune(self.name)
        return False


def tile_shape(H, W, interp=0):
    """(tile_h, tile_w) of the tiled forward for H x W slices (the association of its sum, include/ctpvae_radon.h), or None
    when the geometry is not tiled."""
    th, tw = ctypes.c_int(0), ctypes.c_int(0)
    rc = load().ctpvae_rotate_tile_shape(int(H), int(W), int(interp), ctypes.byref(th), ctypes.byref(tw))
    if rc < 0:
        check(rc, "rotate_tile_shape")
    return (th.value, tw.value) if rc == 1 else None


def last_error():
    return load().ctpvae_last_error().decode("utf-8", "replace")


def check(rc, what):
    """Map a C return code to the reference's error convention: ValueError for bad arguments
    (the reference raises ValueError on shape mismatches, ctvae/fbp_tensorflow.py:43-45), RuntimeError otherwise."""
    if rc >= 0:
        return rc
    msg = f"{what}: {last_error()}"
    if rc == EINVAL:
        raise ValueError(msg)
    raise RadonLibraryError(msg)
