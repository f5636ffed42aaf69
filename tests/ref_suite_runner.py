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


def main(argv) -> int:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from marqo_amd.s2_inference.processing import text as product_text
    from oracle import ref_shim
    ref_shim.install(sent_tokenize=product_text._sentences, word_tokenize=product_text._WORD.findall)
    import marqo.s2_inference  # noqa: F401  (the package itself stays the reference's: only the listed modules are replaced)
    for ref_name, our_name in ALIASES.items():
        ours = importlib.import_module(our_name)
        sys.modules[ref_name] = ours
        parent, _, leaf = ref_name.rpartition(".")
        try:
            setattr(importlib.import_module(parent), leaf, ours)
        except ImportError:
            pass
    import pytest
    files = [a for a in argv if not a.startswith("-")]
    extra = [a for a in argv if a.startswith("-")]
    return int(pytest.main(["-q", "-p", "no:cacheprovider", "--noconftest", "--import-mode=importlib", "--rootdir", "/tmp", "-o", "addopts=", *extra, *files]))


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
