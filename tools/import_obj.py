#!/usr/bin/env python3
"""Convert a Wavefront OBJ control mesh into the repo's compact mesh fixture (.npz).

The only scene asset the benchmarks need is the reference's `build/bomberman.obj` (742 vertices, 727 quads,
a data file, not source).  It is stored here re-encoded as numpy arrays so that tests and bench.py do not
depend on an OBJ parser or on /root/reference at run time:

    verts      float32 [nv,3]   the `v` lines, correctly rounded to fp32
    face_sizes uint32  [nf]     vertices per face
    face_index uint32  [sum]    0-based vertex indices, faces concatenated

The loader semantics follow tutorials/common/scenegraph/obj_loader.cpp:517-591 of the reference: `o`, `g`,
`usemtl`, `s`, `vn`, `vt` lines do not split the mesh (one geometry), negative indices are relative.

    python tools/import_obj.py /root/reference/build/bomberman.obj assets/bomberman.mesh.npz
"""
import sys

import numpy as np


def load_obj(path):
    verts, sizes, index = [], [], []
    with open(path, "r") as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                verts.append([float(tok[1]), float(tok[2]), float(tok[3])])
            elif tok[0] == "f":
                ids = []
                for t in tok[1:]:
                    i = int(t.split("/")[0])
                    ids.append(i - 1 if i > 0 else len(verts) + i)
                sizes.append(len(ids))
                index.extend(ids)
    return (np.asarray(verts, dtype=np.float32), np.asarray(sizes, dtype=np.uint32), np.asarray(index, dtype=np.uint32))


def main():
    src, dst = sys.argv[1], sys.argv[2]
    v, s, i = load_obj(src)
    np.savez_compressed(dst, verts=v, face_sizes=s, face_index=i)
    print(f"{src}: {len(v)} vertices, {len(s)} faces -> {dst}")


if __name__ == "__main__":
    main()
