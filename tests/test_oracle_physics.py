"""Known-answer and invariant tests that pin the CPU oracle's physics (oracle/cosim_oracle.c).

PARITY UNPINNED at the MuJoCo boundary (no reference fixture exists and MuJoCo cannot run here):
these tests pin the restatement through physics that does not depend on MuJoCo — an independent Jacobian-based mass
matrix, gravity bias from first principles, the Coriolis power identity, exact discrete free fall, angular momentum,
static force balance and the KKT conditions of the constraint solve.
"""
import numpy as np
import pytest

from cosim_amd import compile as cc
from cosim_amd.compile import compile_model
from cosim_amd.config import PARITY_RANDOM, make_config
from cosim_amd.model import get_field, set_field
from oracle.oracle import Oracle


@pytest.fixture(scope="module")
def cm():
    return compile_model(make_config("flamingo_light_v1", random=PARITY_RANDOM))


def _random_state(cm, rng, z=1.0, vel=1.0):
    nq, nv = cm.blob.nq, cm.blob.nv
    q = np.array(get_field(cm.blob, "init_qpos")[:nq])
    q[2] = z
    quat = rng.normal(size=4)
    q[3:7] = quat / np.linalg.norm(quat)
    q[7:] = rng.uniform(-0.5, 0.5, size=nq - 7)
    q[7], q[10] = -abs(q[7]), -abs(q[10])     # shoulders inside their range [-0.75, 0]
    v = rng.normal(size=nv) * vel
    return q, v


def _numpy_M(cm, q):
    m = cm.const["m"]
    fk = cc.forward_kinematics(m, q)
    mconst, mb, _, _ = cc._mass_matrix_parts(m, fk)
    return mconst + np.einsum("b,bij->ij", m["body_mass"], mb), fk


def test_mass_matrix_matches_jacobian_formulation(cm):
    rng = np.random.default_rng(1)
    o = Oracle(cm)
    for _ in range(5):
        q, v = _random_state(cm, rng)
        o.reset(q, v)
        o.forward()
        M_np, _ = _numpy_M(cm, q)
        np.testing.assert_allclose(o.M, M_np, rtol=1e-10, atol=1e-12)
        assert np.all(np.linalg.eigvalsh(o.M) > 0)


def test_gravity_bias_from_first_principles(cm):
    rng = np.random.default_rng(2)
    o = Oracle(cm)
    m = cm.const["m"]
    g = np.array(get_field(cm.blob, "gravity"))
    for _ in range(3):
        q, _ = _random_state(cm, rng)
        o.reset(q, np.zeros(cm.blob.nv))
        o.forward()
        fk = cc.forward_kinematics(m, q)
        expect = np.zeros(cm.blob.nv)
        for b in range(1, cm.blob.nbody):
            jp, _ = cc.body_jacobian(m, fk, b, fk["xipos"][b])
            expect -= m["body_mass"][b] * jp.T @ g
        np.testing.assert_allclose(o.qfrc_bias, expect, rtol=1e-10, atol=1e-12)


def _integrate_pos(q, v, h):
    q = q.copy()
    q[0:3] += h * v[0:3]
    w = v[3:6]
    n = np.linalg.norm(w)
    if n > 0:
        dq = cc.axis_angle_quat(w / n, h * n)
        q[3:7] = cc.quat_mul(q[3:7], dq)
    q[7:] += h * v[6:]
    return q


def test_coriolis_power_identity(cm):
    """v . C(q, v) v = 0.5 v . Mdot v  (skew-symmetry of Mdot - 2C), Mdot by central differences along the motion."""
    rng = np.random.default_rng(3)
    o = Oracle(cm)
    for _ in range(3):
        q, v = _random_state(cm, rng)
        o.reset(q, v)
        o.forward()
        bias_v = o.qfrc_bias.copy()
        o.reset(q, np.zeros_like(v))
        o.forward()
        cor = bias_v - o.qfrc_bias
        eps = 1e-6
        Mp, _ = _numpy_M(cm, _integrate_pos(q, v, eps))
        Mm, _ = _numpy_M(cm, _integrate_pos(q, v, -eps))
        Mdot = (Mp - Mm) / (2 * eps)
        assert v @ cor == pytest.approx(0.5 * v @ Mdot @ v, rel=1e-5, abs=1e-7)


def _free_flight_oracle(cm):
    o = Oracle(cm)
    set_field(o.model, "dof_frictionloss", np.zeros(cm.blob.nv))   # internal dissipation off; equalities stay on
    return o


def test_free_fall_com_is_exact_and_momentum_conserved(cm):
    rng = np.random.default_rng(4)
    o = _free_flight_oracle(cm)
    q, v = _random_state(cm, rng, z=3.0, vel=0.5)
    o.reset(q, v)
    o.forward()
    mass = np.array(get_field(cm.blob, "body_mass")[:cm.blob.nbody])
    com0 = o.subtree_com[1].copy()

    def momenta():
        cv, com, xi = o.cvel, o.subtree_com[1], o.xipos
        p, L = np.zeros(3), np.zeros(3)
        xim = o.ximat if hasattr(o, "ximat") else None
        for b in range(1, cm.blob.nbody):
            w = cv[b, :3]
            vb = cv[b, 3:] + np.cross(w, xi[b] - com)
            p += mass[b] * vb
        return p

    h, g = cm.blob.timestep, 9.81
    p0 = momenta()
    K = 100
    for k in range(1, K + 1):
        o.step()
        assert o.ncon == 0
    o.forward()
    com = o.subtree_com[1]
    vz0 = p0[2] / mass.sum()
    # closed form of semi-implicit Euler for a point mass: z_K = z0 + K h vz0 - g h^2 K (K + 1) / 2 ; x, y uniform.
    # The articulated system integrates in joint space, so the CoM follows it up to the O(h^2)-per-step integrator
    # error (the CoM is nonlinear in q): 1e-4 m over a 1.3 m drop; momentum to 1e-4 relative.
    assert com[2] == pytest.approx(com0[2] + K * h * vz0 - g * h * h * K * (K + 1) / 2, abs=1e-4)
    np.testing.assert_allclose(com[:2], com0[:2] + K * h * p0[:2] / mass.sum(), atol=1e-4)
    p1 = momenta()
    np.testing.assert_allclose(p1[:2], p0[:2], atol=2e-3)
    assert p1[2] == pytest.approx(p0[2] - mass.sum() * g * K * h, rel=2e-4)


def test_static_stance_force_balance_and_kkt(cm):
    o = Oracle(cm)
    o.reset(np.array(get_field(cm.blob, "init_qpos")[:cm.blob.nq]))
    for _ in range(300):
        o.control_step(np.zeros(4))
    o.forward()
    assert np.abs(o.qvel).max() < 2e-3
    con = o.contacts()
    f = o.efc_force
    total = sum(f[int(c[8]):int(c[8]) + 4].sum() for c in con if c[8] >= 0)     # sum of pyramid edges = normal force
    weight = np.array(get_field(cm.blob, "body_mass")[:cm.blob.nbody]).sum() * 9.81
    assert total == pytest.approx(weight, rel=1e-6)
    # KKT of the convex problem: gradient vanishes, forces respect their sets
    grad = o.M @ o.qacc - o.qfrc_smooth - o.J.T @ f
    assert np.abs(grad).max() < 1e-6
    ne, nf = o.ne, o.nf
    floss = np.array(get_field(cm.blob, "dof_frictionloss")[:cm.blob.nv])
    assert np.all(np.abs(f[ne:ne + nf]) <= floss[floss > 0] + 1e-12)
    assert np.all(f[ne + nf:] >= -1e-12)
    # closed 4-bar: connect residual stays inside the 1 mm impedance width
    assert np.abs(o.efc_pos[:ne]).max() < 1.2e-3


def test_newton_solution_is_the_minimiser(cm):
    """Perturbing qacc can only increase the convex cost the solver minimises."""
    rng = np.random.default_rng(5)
    o = Oracle(cm)
    o.reset(np.array(get_field(cm.blob, "init_qpos")[:cm.blob.nq]))
    for _ in range(20):
        o.control_step(0.2 * rng.normal(size=4))
    o.forward()

    def cost(a):
        jar = o.J @ a - o.efc_aref
        D, R = o.efc_D, o.efc_R
        c = 0.5 * (a - o.qacc_smooth) @ o.M @ (a - o.qacc_smooth)
        ne, nf = o.ne, o.nf
        fl = np.array(get_field(cm.blob, "dof_frictionloss")[:cm.blob.nv])
        fl = fl[fl > 0]
        for i, x in enumerate(jar):
            if i < ne:
                c += 0.5 * D[i] * x * x
            elif i < ne + nf:
                f_ = fl[i - ne]
                c += 0.5 * D[i] * x * x if abs(x) < R[i] * f_ else f_ * (abs(x) - 0.5 * R[i] * f_)
            elif x < 0:
                c += 0.5 * D[i] * x * x
        return c

    c0 = cost(o.qacc)
    for _ in range(50):
        d = rng.normal(size=cm.blob.nv) * 10 ** rng.uniform(-4, 0)
        assert cost(o.qacc + d) >= c0 - 1e-9 * max(1.0, abs(c0))


def test_mpr_known_answers_on_primitives():
    """libccd-style MPR restatement (oracle/cosim_oracle.c mpr_penetration) on pairs with closed-form answers."""
    import ctypes
    from oracle.oracle import lib
    L = lib()

    def q(k1, p1, R1, s1, k2, p2, R2, s2):
        a = np.concatenate([p1, np.asarray(R1).ravel()]).astype(np.float64)
        b = np.concatenate([p2, np.asarray(R2).ravel()]).astype(np.float64)
        s1 = np.asarray(s1, dtype=np.float64); s2 = np.asarray(s2, dtype=np.float64)
        out = np.zeros(7)
        rc = L.oracle_mpr_prims(k1, a.ctypes.data, s1.ctypes.data, k2, b.ctypes.data, s2.ctypes.data, out.ctypes.data)
        return rc, out[0], out[1:4], out[4:7]
    I = np.eye(3)
    SPH, CYL, BOX = 0, 1, 2
    # two spheres: depth = r1 + r2 - |c|, normal along the centre line, position midway through the overlap
    c = np.array([0.6, 0.5, 0.3])
    rc, depth, n, pos = q(SPH, [0, 0, 0], I, [0.5, 0, 0], SPH, c, I, [0.5, 0, 0])
    assert rc == 0 and depth == pytest.approx(1 - np.linalg.norm(c), abs=1e-6)
    np.testing.assert_allclose(n, c / np.linalg.norm(c), atol=1e-4)
    np.testing.assert_allclose(pos, 0.5 * c, atol=1e-4)
    # axis-aligned boxes overlapping 0.1 along x
    rc, depth, n, pos = q(BOX, [0, 0, 0], I, [0.5, 0.5, 0.5], BOX, [0.9, 0.1, 0.05], I, [0.5, 0.5, 0.5])
    assert rc == 0 and depth == pytest.approx(0.1, abs=1e-9)
    np.testing.assert_allclose(n, [1, 0, 0], atol=1e-9)
    assert pos[0] == pytest.approx(0.45, abs=1e-9)
    # separated boxes -> no contact
    assert q(BOX, [0, 0, 0], I, [0.5, 0.5, 0.5], BOX, [1.1, 0, 0], I, [0.5, 0.5, 0.5])[0] == -1
    # crossed cylinders (axes z and y), centres 0.55 apart along x with radii 0.3: depth 0.05 along x
    Rx = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=float)
    rc, depth, n, pos = q(CYL, [0, 0, 0], I, [0.3, 0.5, 0], CYL, [0.55, 0, 0], Rx, [0.3, 0.5, 0])
    assert rc == 0 and depth == pytest.approx(0.05, abs=1e-5)
    np.testing.assert_allclose(n, [1, 0, 0], atol=1e-3)
    np.testing.assert_allclose(pos, [0.275, 0, 0], atol=1e-3)


def test_self_collision_forces_are_internal():
    """Robot-robot contacts act equal and opposite on the two bodies (mj_jacDifPair / mj_rnePostConstraint): summed over
    the bodies, cfrc_ext holds only the ground reaction, and switching the pairs off changes the motion."""
    from cosim_amd.compile import compile_model
    from cosim_amd.config import PARITY_RANDOM, make_config
    from cosim_amd.model import get_field
    from oracle.oracle import Oracle
    cm = compile_model(make_config("humanoid_p_v0", random=PARITY_RANDOM))
    b = cm.blob
    o = Oracle(cm)
    o.reset(np.array(get_field(b, "init_qpos")[:b.nq]))
    phi = np.random.default_rng(0).uniform(0, 6.28, b.nu)
    seen = 0
    for t in range(300):
        o.control_step(np.clip(0.6 * np.sin(2 * np.pi * 0.5 * t * 0.02 + phi), -1, 1))
        c = o.contacts()
        if len(c) and (c[:, 9] >= 0).any() and (c[:, 8] >= 0).all():
            seen += 1
            f = o.efc_force
            ground = np.zeros(3)
            for ci in c[c[:, 9] < 0]:
                adr = int(ci[8])
                ground += ci[4:7] * f[adr:adr + 4].sum()          # normal part of the pyramid force; tangents cancel in z only
            total = o.cfrc_ext[:b.nbody, 3:6].sum(axis=0)
            assert total[2] == pytest.approx(ground[2], rel=1e-9, abs=1e-9)   # ground normal is +z on the plane
    assert seen >= 10 and not o.bad
    z_with = o.qpos.copy()
    o.reset(np.array(get_field(b, "init_qpos")[:b.nq]))
    o.set_self_collision(False)
    for t in range(300):
        o.control_step(np.clip(0.6 * np.sin(2 * np.pi * 0.5 * t * 0.02 + phi), -1, 1))
    assert np.abs(o.qpos - z_with).max() > 1e-3
