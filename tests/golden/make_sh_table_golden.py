"""
Generates tests/golden/sh_table_v1.npz — the only numeric table the reference holds on the hot path: the 64 real
spherical-harmonic polynomials of spherical_harmonic() (reference learn_nerf/ref_nerf.py:174-311, constants from
tiny-cuda-nn), evaluated in float64 at fixed unit vectors.

The reference module cannot be imported here (JAX is absent), so the file is read as TEXT and parsed with `ast`; only
the arithmetic of the `out[i] = <polynomial in x, y, z>` assignments inside `populate()` is evaluated, by the small
whitelist interpreter below (numbers, the coordinate monomials, + - * and unary minus) — nothing of the reference is
executed or copied; the fixture holds inputs and outputs only.

  python tests/golden/make_sh_table_golden.py   (needs /root/reference; run in the build container, not on the GPU box)
"""
import ast
import os

import numpy as np

REF = "/root/reference/learn_nerf/ref_nerf.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sh_table_v1.npz")


def directions():
    rng = np.random.default_rng(20240607)
    d = rng.normal(size=(10, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    axes = np.array([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, -1.0, 0.0], [0.6, 0.0, -0.8]])
    return np.concatenate([axes, d], axis=0)


def evaluate(node, env):
    if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
        return np.float64(node.value)
    if isinstance(node, ast.Name) and node.id in env:
        return env[node.id]
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
        return -evaluate(node.operand, env)
    if isinstance(node, ast.BinOp) and isinstance(node.op, (ast.Add, ast.Sub, ast.Mult)):
        a, b = evaluate(node.left, env), evaluate(node.right, env)
        return a + b if isinstance(node.op, ast.Add) else (a - b if isinstance(node.op, ast.Sub) else a * b)
    if isinstance(node, ast.Call):  # out[0] = broadcast_to(array(c), x.shape): the constant c
        consts = [n.value for n in ast.walk(node) if isinstance(n, ast.Constant) and isinstance(n.value, float)]
        if len(consts) == 1:
            return np.full_like(env["x"], consts[0])
    raise ValueError(f"unexpected syntax in the polynomial table: {ast.dump(node)[:120]}")


def main():
    tree = ast.parse(open(REF).read())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "spherical_harmonic")
    populate = next(n for n in ast.walk(fn) if isinstance(n, ast.FunctionDef) and n.name == "populate")
    dirs = directions()
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    env = dict(x=x, y=y, z=z, xy=x * y, xz=x * z, yz=y * z, x2=x * x, y2=y * y, z2=z * z)
    env.update(x4=env["x2"] ** 2, y4=env["y2"] ** 2, z4=env["z2"] ** 2)
    env.update(x6=env["x4"] * env["x2"], y6=env["y4"] * env["y2"], z6=env["z4"] * env["z2"])
    values = np.full((dirs.shape[0], 64), np.nan)
    for stmt in populate.body:
        if isinstance(stmt, ast.Assign) and isinstance(stmt.targets[0], ast.Subscript) \
                and getattr(stmt.targets[0].value, "id", None) == "out":
            index = stmt.targets[0].slice.value
            values[:, index] = evaluate(stmt.value, env)
    assert not np.isnan(values).any(), "every one of the 64 basis functions must have been found"
    np.savez(OUT, dirs=dirs, values=values,
             source=np.array("reference learn_nerf/ref_nerf.py:174-311 polynomial table, float64"))
    print(f"wrote {OUT}: {values.shape}")


if __name__ == "__main__":
    main()
