"""Run the REFERENCE'S OWN unit-test files for the vectorise() plumbing against the PRODUCT's modules.

In a fresh interpreter: oracle/ref_shim.py makes /root/reference/src importable (stubs for the wheels this image lacks), then the module
names the reference's tests import and patch — `marqo.s2_inference.s2_inference`, `marqo.s2_inference.random_utils`, ... — are bound to
marqo_amd's modules in sys.modules, so `from marqo.s2_inference import s2_inference` and `mock.patch('marqo.s2_inference.s2_inference._load_model')`
inside those test files reach marqo_amd.s2_inference.s2_inference.  Then pytest runs the reference's test files where they lie (nothing is
copied).  usage: python tests/ref_suite_runner.py <reference test file> [...] [-- pytest args]; prints pytest's own summary.
Test infrastructure only (tests/test_ref_parity.py::test_reference_unit_tests_pass_on_the_product drives it)."""
import importlib
import os
import sys

ALIASES = {   # reference module name -> product module that stands in for it
    "marqo.s2_inference.s2_inference": "marqo_amd.s2_inference.s2_inference",
    "marqo.s2_inference.random_utils": "marqo_amd.s2_inference.random_utils",
    "marqo.s2_inference.model_registry": "marqo_amd.s2_inference.model_registry",
    "marqo.inference.inference_cache.marqo_inference_cache": "marqo_amd.s2_inference.inference_cache",
    "marqo.inference.inference_cache.marqo_lru_cache": "marqo_amd.s2_inference.inference_cache",
    "marqo.inference.inference_cache.marqo_lfu_cache": "marqo_amd.s2_inference.inference_cache",
    "marqo.s2_inference.processing.text": "marqo_amd.s2_inference.processing.text",
    "marqo.s2_inference.sbert_utils": "marqo_amd.s2_inference.sbert_utils",
}


def _provide_test_constants() -> None:
    """Several reference test files do `from tests.marqo_test import TestImageUrls` — an enum of image URLs that sits in a module which
    also starts a Vespa client on import.  Read just that class out of the reference's file (ast, nothing executed but its constant
    assignments) and offer it as `tests.marqo_test`."""
    import ast
    import enum
    import types
    from oracle import ref_shim
    path = os.path.join(os.path.dirname(ref_shim.REFERENCE_SRC), "tests", "marqo_test.py")
    members = {}
    with open(path, encoding="utf-8") as f:
        for node in ast.parse(f.read()).body:
            if isinstance(node, ast.ClassDef) and node.name == "TestImageUrls":
                for stmt in node.body:
                    if isinstance(stmt, ast.Assign) and isinstance(stmt.value, ast.Constant) and isinstance(stmt.value.value, str):
                        members[stmt.targets[0].id] = stmt.value.value
    pkg = types.ModuleType("tests")
    pkg.__path__ = []
    mod = types.ModuleType("tests.marqo_test")
    mod.TestImageUrls = enum.Enum("TestImageUrls", members, type=str)
    mod.TestImageUrls.__test__ = False
    pkg.marqo_test = mod
    sys.modules["tests"], sys.modules["tests.marqo_test"] = pkg, mod


def main(argv) -> int:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import ref_shim, segment
    # (punkt is not downloadable: the reference's own code paths that still call nltk get the independent segmenters of oracle/segment.py)
    ref_shim.install(sent_tokenize=segment.sentences, word_tokenize=segment.words)
    import marqo.s2_inference  # noqa: F401  (the package itself stays the reference's: only the listed modules are replaced)
    for ref_name, our_name in ALIASES.items():
        ours = importlib.import_module(our_name)
        sys.modules[ref_name] = ours
        parent, _, leaf = ref_name.rpartition(".")
        try:
            setattr(importlib.import_module(parent), leaf, ours)
        except ImportError:
            pass
    _provide_test_constants()
    import pytest
    files = [a for a in argv if a.endswith(".py")]
    extra = [a for a in argv if not a.endswith(".py")]
    return int(pytest.main(["-q", "-p", "no:cacheprovider", "--noconftest", "--import-mode=importlib", "--rootdir", "/tmp", "-o", "addopts=", *extra, *files]))


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
