#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the read-only reference checkout.

Run in the BUILD container only (needs /root/reference); the GPU box never runs it.
  python tests/golden/make_golden.py

go_vectorstore.json — the Go provider conformance fixture
  (provider/test_files/embeddings.csv: 5 entities x 768 floats, and the search vector
  hard-coded at provider/vectorstore_test.go:214-226), parsed exactly as the Go test
  does (strconv.ParseFloat(..., 32) == round-to-nearest fp32), plus the float64 numpy
  cosine/L2 distances used as an implementation-independent cross-check of the
  expected `Nearest(..., k=2)` answer (SURVEY.md §8c "Go boundary").
"""
import csv
import json
import os
import re

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    with open(os.path.join(REF, "provider/test_files/embeddings.csv")) as f:
        rows = list(csv.reader(f))[1:]
    entities = [r[0] for r in rows]
    vecs = [[float(np.float32(float(x))) for x in r[1].split(",")] for r in rows]
    src = open(os.path.join(REF, "provider/vectorstore_test.go")).read()
    m = re.search(r'func getSearchVector.*?vectorStr := "([^"]+)"', src, re.S)
    query = [float(np.float32(float(x))) for x in m.group(1).split(",")]
    X = np.array(vecs, dtype=np.float64)
    q = np.array(query, dtype=np.float64)
    cos = 1.0 - (X @ q) / (np.linalg.norm(X, axis=1) * np.linalg.norm(q))
    l2 = ((X - q) ** 2).sum(1)
    order = [entities[i] for i in np.argsort(cos)]
    json.dump({
        "source": ["provider/test_files/embeddings.csv", "provider/vectorstore_test.go:214-226"],
        "entities": entities, "vectors": vecs, "query": query,
        "float64_cosine_distance": cos.tolist(), "float64_l2sq_distance": l2.tolist(),
        "nearest_cosine_order": order,
    }, open(os.path.join(OUT, "go_vectorstore.json"), "w"))
    print("wrote go_vectorstore.json", order, cos)


if __name__ == "__main__":
    main()
