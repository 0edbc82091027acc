"""Binary-PLY checkpoint interchange with the reference (GaussianModel::savePly / loadPly,
src/gaussian_model.cpp:838-1047) and with Inria 3DGS viewers: one `vertex` element of float32 properties
  x y z  nx ny nz  f_dc_0..2  f_rest_0..(3*(M-1)-1)  opacity  scale_0..2  rot_0..3
holding the RAW (pre-activation) parameters; f_dc / f_rest are channel-major (features.transpose(1, 2).flatten(1))."""
import numpy as np


def _names(n_rest):
    return (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(n_rest)] +
            ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])


def save_ply(path, xyz, features, opacity, scaling, rotation):
    """features: [P, M, 3] (dc first).  All arrays float32 numpy (raw parameters)."""
    P, M = features.shape[0], features.shape[1]
    f_dc = np.transpose(features[:, :1, :], (0, 2, 1)).reshape(P, 3)
    f_rest = np.transpose(features[:, 1:, :], (0, 2, 1)).reshape(P, 3 * (M - 1))
    rows = np.concatenate([xyz, np.zeros_like(xyz), f_dc, f_rest, opacity.reshape(P, 1), scaling, rotation], axis=1)
    names = _names(3 * (M - 1))
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {P}\n" + \
        "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(rows, dtype="<f4").tobytes())


def load_ply(path, max_sh_degree=3):
    """Returns dict(xyz, features [P,M,3], opacity [P,1], scaling, rotation) of float32 arrays."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt = f.readline().split()
        if fmt[:2] != [b"format", b"binary_little_endian"]:
            raise ValueError("only binary_little_endian PLY is supported (what savePly writes)")
        count, props = 0, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError("unterminated PLY header")
            tok = line.split()
            if tok[:1] == [b"end_header"]:
                break
            if tok[:2] == [b"element", b"vertex"]:
                count = int(tok[2])
            elif tok[:1] == [b"property"]:
                if tok[1] not in (b"float", b"float32"):
                    raise ValueError("unexpected property type " + tok[1].decode())
                props.append(tok[2].decode())
        data = np.frombuffer(f.read(count * len(props) * 4), dtype="<f4").reshape(count, len(props))
    col = {n: i for i, n in enumerate(props)}
    take = lambda names: np.stack([data[:, col[n]] for n in names], 1)
    M = (max_sh_degree + 1) ** 2
    n_rest = 3 * (M - 1)
    f_dc = take([f"f_dc_{i}" for i in range(3)]).reshape(count, 3, 1)
    f_rest = take([f"f_rest_{i}" for i in range(n_rest)]).reshape(count, 3, M - 1)
    features = np.concatenate([np.transpose(f_dc, (0, 2, 1)), np.transpose(f_rest, (0, 2, 1))], axis=1)
    return dict(xyz=take(["x", "y", "z"]).copy(), features=np.ascontiguousarray(features, np.float32),
                opacity=take(["opacity"]).copy(), scaling=take([f"scale_{i}" for i in range(3)]).copy(),
                rotation=take([f"rot_{i}" for i in range(4)]).copy())
