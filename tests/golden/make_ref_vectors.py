"""Lifts the golden embedding vectors the REFERENCE'S OWN TESTS hold for this path into a fixture (tests/golden/ref_vectors.npz):

  /root/reference/tests/core/inference/embedding_models/test_hugging_face_model.py
      :15-275   E5_BASE_V2_MODEL_EMBEDDINGS          intfloat/e5-base-v2, mean pooling, 'query: how much protein should a female eat' (:614-634)
      :276-535  NLI_BERT_BASE_CLS_MODEL_EMBEDDINGS   sentence-transformers/nli-bert-base-cls-pooling, CLS pooling, 'This is an example sentence' (:742-770)
  /root/reference/tests/core/inference/test_marqo_fashion_clip.py
      :26-557   FASHIONCLIP_{IMAGE,TEXT}_EMBEDDING, SiGLIP_{IMAGE,TEXT}_EMBEDDING   Marqo/marqo-fashionCLIP (ViT-B-16) and
                Marqo/marqo-fashionSigLIP (ViT-B-16-SigLIP) on the fashion-hippo image and the text 'a hat' (:565-615)

The class-level numpy literals are read with `ast` (the reference test modules are not imported: they need the real wheels), so the
numbers are the reference's, digit for digit.  These vectors need the real checkpoints, which do not exist offline:
tests/test_ref_vectors.py runs them whenever `$MARQO_AMD_MODEL_DIR/<repo>` is mounted and skips otherwise.

    python tests/golden/make_ref_vectors.py
"""
import ast
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_TESTS = os.environ.get("MARQO_REFERENCE_TESTS", "/root/reference/tests")

SOURCES = {
    "core/inference/embedding_models/test_hugging_face_model.py": {
        "E5_BASE_V2_MODEL_EMBEDDINGS": dict(model="hf/e5-base-v2", repo="intfloat/e5-base-v2", kind="text", pooling="mean",
                                            content="query: how much protein should a female eat", tol="norm(emb - gold) / 1 < 1e-4"),
        "NLI_BERT_BASE_CLS_MODEL_EMBEDDINGS": dict(model=None, repo="sentence-transformers/nli-bert-base-cls-pooling", kind="text", pooling="cls",
                                                   content="This is an example sentence", tol="norm(emb - gold) / 1 < 1e-4"),
    },
    "core/inference/test_marqo_fashion_clip.py": {
        "FASHIONCLIP_IMAGE_EMBEDDING": dict(model="Marqo/marqo-fashionCLIP", repo="Marqo/marqo-fashionCLIP", kind="image", content="fashion-hippo.png",
                                            tol="norm(emb - gold) / D < 1e-4"),
        "FASHIONCLIP_TEXT_EMBEDDING": dict(model="Marqo/marqo-fashionCLIP", repo="Marqo/marqo-fashionCLIP", kind="text", content="a hat",
                                           tol="norm(emb - gold) / D < 1e-4"),
        "SiGLIP_IMAGE_EMBEDDING": dict(model="Marqo/marqo-fashionSigLIP", repo="Marqo/marqo-fashionSigLIP", kind="image", content="fashion-hippo.png",
                                       tol="norm(emb - gold) / D < 1e-4"),
        "SiGLIP_TEXT_EMBEDDING": dict(model="Marqo/marqo-fashionSigLIP", repo="Marqo/marqo-fashionSigLIP", kind="text", content="a hat",
                                      tol="norm(emb - gold) / D < 1e-4"),
    },
}


def main() -> None:
    arrays, meta = {}, {}
    for rel, wanted in SOURCES.items():
        path = os.path.join(REF_TESTS, rel)
        tree = ast.parse(open(path, encoding="utf-8").read())
        for node in ast.walk(tree):
            if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                    and node.targets[0].id in wanted:
                name = node.targets[0].id
                value = eval(compile(ast.Expression(node.value), path, "eval"), {"np": np, "__builtins__": {}})  # noqa: S307 - numpy literal
                arrays[name] = np.asarray(value, dtype=np.float64)
                meta[name] = dict(wanted[name], source=f"tests/{rel}:{node.lineno}-{node.end_lineno}", dim=int(arrays[name].shape[-1]))
        missing = set(wanted) - set(arrays)
        assert not missing, f"{rel}: not found {missing}"
    np.savez_compressed(os.path.join(HERE, "ref_vectors.npz"), __meta__=np.frombuffer(json.dumps(meta, sort_keys=True).encode(), dtype=np.uint8),
                        **arrays)
    for k, v in arrays.items():
        print(f"{k}: shape {v.shape}, |v| = {np.linalg.norm(v):.6f}  ({meta[k]['source']})")


if __name__ == "__main__":
    main()
