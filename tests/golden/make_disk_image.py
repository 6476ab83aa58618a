"""Hand-assembled on-disk index in the reference's directory format, written WITHOUT this repo's code (struct + numpy only), so that
the loaders are tested against bytes their own writers did not produce.

Format, from the reference's own description (src/cpp/src/dynamic_inverted_list.cpp:339-357, quake_index.cpp:170-205):
    <dir>/metadata.txt   four lines: metric=<faiss code: 1 L2, 0 IP>, level=<n>, ntotal=<n>, nlist=<n>
    <dir>/partitions     32-byte header  [ magic u32 = 0x44494E4C | version u32 = 3 | nlist u64 | code_size u64 | num_partitions u64 ]
                         offsets  u64[num_partitions + 1]   (relative to the start of the chunks)
                         partition ids u64[num_partitions]  (file order = unordered_map order upstream: NOT sorted, :381)
                         chunks: for partition i  [ codes (n_i * code_size bytes: n_i rows of d float32) | ids (n_i int64) ]
    <dir>/parent/...     the same for the flat index over the centroids (one partition, id 0; vector ids = partition ids)

    python tests/golden/make_disk_image.py            # rewrites tests/golden/disk_image/ and disk_image.json (the expected content)
"""
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAGIC, VERSION = 0x44494E4C, 3


def write_partitions(path, d, parts):
    """parts: list of (partition id, vectors [n, d] float32, ids [n] int64) in FILE order"""
    code_size = 4 * d
    offsets = [0]
    for _, v, _ in parts:
        offsets.append(offsets[-1] + v.shape[0] * (code_size + 8))
    with open(path, "wb") as f:
        f.write(struct.pack("<IIQQQ", MAGIC, VERSION, len(parts), code_size, len(parts)))
        f.write(struct.pack("<%dQ" % len(offsets), *offsets))
        f.write(struct.pack("<%dQ" % len(parts), *[p for p, _, _ in parts]))
        for _, v, i in parts:
            f.write(np.ascontiguousarray(v, "<f4").tobytes())
            f.write(np.ascontiguousarray(i, "<i8").tobytes())


def main():
    d = 6
    rng = np.random.default_rng(20250930)
    # three partitions in non-ascending file order, one of them empty; small integers so that every distance is exact in fp32
    cent = {5: np.full(d, 10.0, np.float32), 0: np.zeros(d, np.float32), 9: np.full(d, -10.0, np.float32)}
    sizes = {5: 7, 0: 0, 9: 4}
    parts, nxt = [], 100
    for pid in (5, 0, 9):
        n = sizes[pid]
        v = (cent[pid][None, :] + rng.integers(-2, 3, size=(n, d))).astype(np.float32)
        ids = np.arange(nxt, nxt + n, dtype=np.int64)[::-1].copy()  # descending ids: row order is the file's, not the ids'
        nxt += n
        parts.append((pid, v, ids))
    out = os.path.join(HERE, "disk_image")
    os.makedirs(os.path.join(out, "parent"), exist_ok=True)
    ntotal = sum(sizes.values())
    with open(os.path.join(out, "metadata.txt"), "w") as f:
        f.write("metric=1\nlevel=0\nntotal=%d\nnlist=3\n" % ntotal)
    write_partitions(os.path.join(out, "partitions"), d, parts)
    # parent: flat index, one partition (id 0) holding the centroids, vector ids = partition ids, file order 9, 5, 0
    pc = np.stack([cent[9], cent[5], cent[0]])
    with open(os.path.join(out, "parent", "metadata.txt"), "w") as f:
        f.write("metric=1\nlevel=1\nntotal=3\nnlist=1\n")
    write_partitions(os.path.join(out, "parent", "partitions"), d, [(0, pc, np.array([9, 5, 0], np.int64))])
    expect = {"d": d, "metric": "l2", "ntotal": ntotal, "nlist": 3,
              "partitions": {str(p): {"ids": i.tolist(), "vectors": v.tolist()} for p, v, i in parts},
              "centroids": {"9": cent[9].tolist(), "5": cent[5].tolist(), "0": cent[0].tolist()}}
    with open(os.path.join(HERE, "disk_image.json"), "w") as f:
        json.dump(expect, f)
    print("wrote", out)


if __name__ == "__main__":
    main()
