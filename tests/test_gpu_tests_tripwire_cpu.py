"""CPU tripwire for the GPU tier: GPUTEST_r04 went red because a GPU test read an attribute (`grad_fn.ws`) a late product
commit had removed, and `-x` then cost 469 tests of evidence.  The GPU tests cannot run here, but the NAMES they reach for can be checked:
every `osvos_*` symbol they call must be exported by the built library, every `module.attr` on a product module they import must exist, every
`from <product module> import name` must resolve, every attribute read off a `grad_fn` must be autograd's own, and every environment switch
they set must still be read by the product."""
import ast
import glob
import importlib
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GPU_FILES = sorted(glob.glob(os.path.join(REPO, "tests", "test_gpu_*.py")) + [os.path.join(REPO, "tests", "test_augment.py"),
                                                                               os.path.join(REPO, "tests", "trained_fixture.py")])
PRODUCT_PREFIXES = ("osvos_pytorch_amd", "networks", "layers", "mypath", "util", "oracle", "bench", "train_online", "train_parent", "golden_util",
                    "trained_fixture")
GRAD_FN_OK = {"saved_tensors", "next_functions", "name", "metadata", "register_hook", "needs_input_grad"}


def _scan(path):
    tree = ast.parse(open(path).read(), path)
    mod_alias, from_imports, osvos_syms, grad_fn_attrs, env_keys = {}, [], set(), set(), set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                if a.name.split(".")[0] in PRODUCT_PREFIXES:
                    mod_alias[a.asname or a.name.split(".")[0]] = a.name if a.asname else a.name.split(".")[0]
        elif isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] in PRODUCT_PREFIXES:
            for a in node.names:
                from_imports.append((node.module, a.name, a.asname or a.name))
        elif isinstance(node, ast.Attribute):
            if node.attr.startswith("osvos_"):
                osvos_syms.add(node.attr)
            if isinstance(node.value, ast.Attribute) and node.value.attr == "grad_fn":
                grad_fn_attrs.add(node.attr)
        elif isinstance(node, ast.Constant) and isinstance(node.value, str) and re.fullmatch(r"OSVOS_[A-Z0-9_]+", node.value):
            env_keys.add(node.value)
    return tree, mod_alias, from_imports, osvos_syms, grad_fn_attrs, env_keys


def test_files_found():
    assert len(GPU_FILES) >= 6


@pytest.mark.parametrize("path", GPU_FILES, ids=[os.path.basename(p) for p in GPU_FILES])
def test_names_the_gpu_tests_reach_for_exist(path):
    from osvos_pytorch_amd import _lib
    l = _lib.lib()
    tree, mod_alias, from_imports, osvos_syms, grad_fn_attrs, env_keys = _scan(path)
    for sym in sorted(osvos_syms):
        assert hasattr(l, sym), "%s calls %s, which libosvos_hip.so does not export" % (os.path.basename(path), sym)
        assert sym in _lib.PROTOTYPES, "%s calls %s without a ctypes prototype" % (os.path.basename(path), sym)
    assert grad_fn_attrs <= GRAD_FN_OK, "%s reads grad_fn.%s" % (os.path.basename(path), sorted(grad_fn_attrs - GRAD_FN_OK))
    modules = {}
    for module, name, asname in from_imports:
        m = importlib.import_module(module)
        if hasattr(m, name):
            obj = getattr(m, name)
        else:
            obj = importlib.import_module(module + "." + name)      # `from package import submodule`
        if isinstance(obj, type(os)):
            modules[asname] = obj
    for alias, name in mod_alias.items():
        modules[alias] = importlib.import_module(name)
    # attribute reads directly on an imported product module: ops.conv3x3(...), _lib.check(...), synth.make_frame(...)
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in modules:
            m = modules[node.value.id]
            if hasattr(m, node.attr):
                continue
            try:
                importlib.import_module(m.__name__ + "." + node.attr)
            except ImportError:
                raise AssertionError("%s: %s.%s does not exist" % (os.path.basename(path), m.__name__, node.attr))
    # every OSVOS_* environment switch a GPU test sets or reads is still read somewhere in the product / bench / scripts
    if env_keys:
        src = ""
        for pat in ("osvos-pytorch_amd/*.py", "osvos-pytorch_amd/*/*.py", "osvos-pytorch_amd/csrc/*", "bench.py", "train_online.py", "train_parent.py",
                    "tools/*.py", "tests/trained_fixture.py"):
            for f in glob.glob(os.path.join(REPO, pat)):
                if os.path.isfile(f) and not f.endswith((".o", ".so")):
                    src += open(f, errors="ignore").read()
        for k in sorted(env_keys):
            if k.startswith("OSVOS_TEST_"):      # read by the tests themselves
                continue
            assert k in src, "%s uses the switch %s, which nothing reads any more" % (os.path.basename(path), k)
