"""Closed-form known answers for the oracle's constraint model on hand-built one-body models (tests/tiny_models.py).

What pins what (MuJoCo is absent; the formulas below are the PUBLISHED constraint model of MuJoCo's "Computation"
chapter, not the oracle's code):
  * impedance interpolation  a1 = (1 - d) a0 + d aref  for a row whose A = J M^-1 J^T is exactly diagApprox (single free body /
    single hinge): pins R = (1 - d)/d * diagApprox, aref = -B v - K d r, (K, B) from solref with dmax = solimp[1], the solimp
    curve (d0, d1, width, midpoint, power) and the Newton solve -- connect rows incl. flamingo_light_v1's solimp="0.001 1";
  * Huber friction-loss row: saturation force exactly frictionloss, creep velocity tau R / B below it;
  * one-sided limit row: rest angle from  tau_gravity = D K d |r|;
  * pyramidal cone geometry and friction mixing: slip on an incline exactly at tan(theta) = max(mu_geom, mu_ground), sliding
    acceleration g sin(theta) - mu g cos(theta);
  * pyramidal regulariser (diagApprox = tran (1 + mu^2), Rpy = 2 mu^2 R): STILL FLAGGED "from memory of engine_core_constraint.c" --
    the rest-penetration test fixes the observable consequence (one number per parameter set) for a future MuJoCo cross-check and
    would fail under the plausible alternatives (Rpy = mu^2 R, no (1 + mu^2) factor, R of the normal row alone).
"""
import numpy as np
import pytest

from oracle.oracle import Oracle
from tiny_models import DEFAULT_SOLIMP, free_sphere, hinge_arm


def imp_curve(solimp, r):
    """d(r) of the MuJoCo documentation (solimp = d0, dwidth, width, midpoint, power), margin 0."""
    d0, d1, w, mid, p = solimp
    d0, d1, mid = (min(0.9999, max(0.0001, v)) for v in (d0, d1, mid))
    x = abs(r) / w
    if x >= 1:
        return d1
    if x <= 0:
        return d0
    y = (x ** p) / (mid ** (p - 1)) if x <= mid else 1 - ((1 - x) ** p) / ((1 - mid) ** (p - 1))
    return d0 + y * (d1 - d0)


def kb(solref, solimp, h):
    d1 = min(0.9999, max(0.0001, solimp[1]))
    tc = max(solref[0], 2 * h)
    return 1.0 / (d1 * d1 * tc * tc * solref[1] ** 2), 2.0 / (d1 * tc)


@pytest.mark.parametrize("solimp", [DEFAULT_SOLIMP, (0.001, 1.0, 0.001, 0.5, 2.0), (0.5, 0.99, 0.002, 0.3, 3.0)])
@pytest.mark.parametrize("delta", [0.0, 0.0005, 0.001, 0.004])
def test_connect_row_interpolates_between_free_and_reference_acceleration(solimp, delta):
    """a1 = (1 - d) a0 + d aref per connect row; second parametrisation = flamingo_light_v1.xml:262-265 (solimp="0.001 1": d = 0.001
    at zero violation, 0.9999 from 1 mm on)."""
    h, g, m = 0.005, 9.81, 2.0
    t = free_sphere(mass=m, radius=0.1, collide=False, connect_to_world=(-delta, 0.0, 0.1), eq_solimp=solimp, timestep=h)
    o = Oracle(t)
    v0 = np.array([0.03, 0.0, -0.02, 0, 0, 0])
    o.reset(None, v0)
    o.forward()
    assert o.nefc == 3 and o.ne == 3
    K, B = kb((0.02, 1.0), solimp, h)
    exp = np.zeros(3)
    a0 = np.array([0.0, 0.0, -g])
    for k, r in enumerate((delta, 0.0, 0.0)):
        d = imp_curve(solimp, r)
        aref = -B * v0[k] - K * d * r
        exp[k] = (1 - d) * a0[k] + d * aref
    np.testing.assert_allclose(o.qacc[:3], exp, rtol=1e-9, atol=1e-9)
    if solimp[0] == 0.001:
        d = [imp_curve(solimp, r) for r in (0.0, 0.0005, 0.001)]
        assert d[0] == 0.001 and d[2] == 0.9999 and d[1] == pytest.approx(0.001 + 0.5 * (0.9999 - 0.001))


def test_friction_loss_row_saturates_at_frictionloss_and_creeps_below_it():
    fl, h = 0.6, 0.002
    t = hinge_arm(gravity=(0, 0, 0), frictionloss=fl, motor=True, timestep=h)
    I = t.inertia_axis
    d0, d1 = DEFAULT_SOLIMP[0], DEFAULT_SOLIMP[1]
    R = (1 - d0) / d0 / I                                     # friction rows sit at r = 0: d = d0; diagApprox = dof_invweight0 = 1/I
    _, B = kb((0.02, 1.0), DEFAULT_SOLIMP, h)
    o = Oracle(t)
    # above the threshold: the row's force is exactly -frictionloss, from the first step on
    tau = 1.0
    o.reset(None, None)
    o.ctrl[0] = tau
    o.forward()
    assert o.nf == 1 and o.nefc == 1
    assert o.qacc[0] == pytest.approx((tau - fl) / I, rel=1e-10) and o.efc_force[0] == pytest.approx(-fl, rel=1e-10)
    assert (tau - fl) / I > R * fl                            # the linear zone of the Huber cost is where the solution lies
    # below it: quadratic zone, force = -D (a + B v); steady creep v = tau R / B (a soft constraint, not a hard stop)
    tau = 0.25
    o.reset(None, None)
    o.ctrl[0] = tau
    for _ in range(4000):
        o.step()
    assert abs(o.qacc[0]) < 1e-9 * tau / I
    assert o.qvel[0] == pytest.approx(tau * R / B, rel=1e-6)
    assert o.efc_force[0] == pytest.approx(-tau, rel=1e-8)


def test_limit_row_rest_angle_under_gravity():
    h, g, m, L = 0.002, 9.81, 1.5, 0.4
    hi = 0.3
    t = hinge_arm(mass=m, length=L, limited=True, jrange=(-0.5, hi), timestep=h, damping=0.0)
    I = t.inertia_axis
    o = Oracle(t)
    o.reset(np.array([hi - 0.01]), None)
    for _ in range(3000):
        o.step()
    assert o.nl == 1 and abs(o.qvel[0]) < 1e-10
    viol = o.qpos[0] - hi
    K, _ = kb((0.02, 1.0), DEFAULT_SOLIMP, h)
    # balance: m g L cos(q) = f,  f = D (aref - 0) = d / ((1 - d) / I) * K d viol   (solved for viol by fixed point)
    v = 1e-4
    for _ in range(200):
        d = imp_curve(DEFAULT_SOLIMP, v)
        v = m * g * L * np.cos(hi + v) * (1 - d) / (I * K * d * d)
    assert 1e-5 < v < 1e-3 and viol == pytest.approx(v, rel=1e-6)
    assert o.efc_force[0] == pytest.approx(m * g * L * np.cos(o.qpos[0]), rel=1e-8) and o.efc_force[0] > 0


@pytest.mark.parametrize("mu_geom,mu_ground", [(0.5, 0.8), (0.8, 0.3), (0.05, 0.8)])
def test_pyramidal_cone_slips_exactly_at_the_larger_friction_coefficient(mu_geom, mu_ground):
    """Incline by tilting gravity.  The contact's friction is max(geom, ground) (equal priorities; the 0.05 / 0.8 pair is the caster
    sphere on the ground of flamingo_light_v1.xml:166).  Below tan(theta) = mu the sphere only creeps; above it the
    loaded pyramid edge is n - mu x: a_x + mu a_z = g (sin(theta) - mu cos(theta))."""
    mu, g, h, m = max(mu_geom, mu_ground), 9.81, 0.002, 2.0
    for frac, slides in ((0.9, False), (1.25, True)):
        th = np.arctan(frac * mu)
        t = free_sphere(mass=m, radius=0.1, inertia=1e5, mu=mu_geom, ground_mu=mu_ground, timestep=h,
                        gravity=(g * np.sin(th), 0.0, -g * np.cos(th)))
        o = Oracle(t)
        o.reset(None, None)
        if slides:
            # A fast-sliding soft pyramidal contact chatters (the -B v term of the loaded edge keeps pushing the sphere out), so
            # there is no steady normal force to quote; what holds in EVERY step in which a single edge n - mu x carries the load
            # f is  m a_x = m g sin - mu f,  m a_z = -m g cos + f,  i.e.  a_x + mu a_z = g (sin(theta) - mu cos(theta))  exactly.
            checked = 0
            for _ in range(60):
                o.step()
                if o.ncon == 1 and (o.efc_force[:4] > 1e-9).sum() == 1:
                    assert o.qacc[0] + mu * o.qacc[2] == pytest.approx(g * (np.sin(th) - mu * np.cos(th)), rel=1e-9)
                    checked += 1
            assert checked >= 10 and o.qvel[0] > 0.05                 # it does slide
        else:
            for _ in range(300):
                o.step()
            assert o.ncon == 1 and o.nefc == 4
            assert abs(o.qacc[0]) < 1e-4 and 0 < o.qvel[0] < 0.02     # sticks (soft constraint: slow creep down the slope)
            f = o.efc_force[:4]
            assert f.sum() == pytest.approx(m * g * np.cos(th), rel=1e-5)                         # N
            assert mu * abs(f[2] - f[3]) + mu * abs(f[0] - f[1]) == pytest.approx(m * g * np.sin(th), rel=1e-4)   # friction holds it


@pytest.mark.parametrize("mu,impratio", [(1.0, 1.0), (0.8, 1.0), (0.5, 4.0)])
def test_sphere_rest_penetration_documents_the_pyramidal_regulariser(mu, impratio):
    """FLAGGED (restated from memory of mj_instantiateContact / mj_makeImpedance): edge rows share diagApprox = (1/m)(1 + mu^2) and
    R = 2 (mu^2 / impratio) (1 - d)/d diagApprox.  At rest sum_e f_e = 4 D K d pen = m g, hence
        pen = g (mu^2 / impratio) (1 + mu^2) (1 - d) / (2 K d^2),     d = d(pen),
    independent of the mass.  Any of the plausible alternatives changes this number by a factor (1/2, 1/(1 + mu^2), ...)."""
    g, h = 9.81, 0.002
    pens = []
    for m in (0.7, 5.0):
        t = free_sphere(mass=m, radius=0.1, mu=mu, ground_mu=mu, timestep=h)
        t.blob.impratio = impratio
        o = Oracle(t)
        o.reset(None, None)
        for _ in range(3000):
            o.step()
        assert o.ncon == 1 and abs(o.qvel[2]) < 1e-12
        pens.append(0.1 - o.qpos[2])
    K, _ = kb((0.02, 1.0), DEFAULT_SOLIMP, h)
    p = 1e-4
    for _ in range(200):
        d = imp_curve(DEFAULT_SOLIMP, p)
        p = g * (mu * mu / impratio) * (1 + mu * mu) * (1 - d) / (2 * K * d * d)
    assert pens[0] == pytest.approx(p, rel=1e-6) and pens[1] == pytest.approx(p, rel=1e-6)
    for alt in (0.5, 2.0, 1.0 / (1 + mu * mu)):                       # what the alternatives would predict (to first order in d)
        assert abs(pens[0] / (p * alt) - 1) > 0.2 or alt == 1.0


def test_box_box_manifolds_have_closed_form_vertices():
    """mjc_BoxBox restatement (oracle box_box_points; MuJoCo's collision table sends box-box pairs there, not to MPR).  The clipped
    incident face is elementary geometry: a small box resting on a large one gives its four bottom corners, two equal boxes turned by
    45 degrees the octagon |x| = h or |y| = h with |x| + |y| = sqrt(2) h, crossed edges one point midway between the edges; contact
    positions lie midway between the two surfaces and dist = -penetration; `margin` admits separated faces."""
    from oracle.oracle import box_box
    I = np.eye(3)
    pts, dist, n = box_box([0, 0, 0], I, [0.5, 0.5, 0.1], [0.1, 0.05, 0.199], I, [0.1, 0.1, 0.1])
    assert len(pts) == 4 and np.allclose(dist, -0.001) and np.allclose(n, [0, 0, 1])
    assert {(round(p[0], 9), round(p[1], 9)) for p in pts} == {(0.0, -0.05), (0.2, -0.05), (0.2, 0.15), (0.0, 0.15)}
    assert np.allclose(pts[:, 2], 0.0995)
    # the large box as geom 2: same points, normal still from box 1 to box 2
    pts2, dist2, n2 = box_box([0.1, 0.05, 0.199], I, [0.1, 0.1, 0.1], [0, 0, 0], I, [0.5, 0.5, 0.1])
    assert len(pts2) == 4 and np.allclose(dist2, -0.001) and np.allclose(n2, [0, 0, -1]) and np.allclose(pts2[:, 2], 0.0995)
    c = np.cos(np.pi / 4)
    Rz = np.array([[c, -c, 0], [c, c, 0], [0, 0, 1]])
    h = 0.1
    pts, dist, n = box_box([0, 0, 0], I, [h, h, h], [0, 0, 2 * h - 0.002], Rz, [h, h, h])
    assert len(pts) == 8 and np.allclose(dist, -0.002) and np.allclose(pts[:, 2], h - 0.001)
    a = np.abs(pts[:, :2])
    assert np.allclose(a.max(axis=1), h) and np.allclose(a.sum(axis=1), np.sqrt(2) * h)
    assert len({(round(p[0], 6), round(p[1], 6)) for p in pts}) == 8
    Rx = np.array([[1, 0, 0], [0, c, -c], [0, c, c]])
    Ry = np.array([[c, 0, c], [0, 1, 0], [-c, 0, c]])
    e = h * np.sqrt(2)
    pts, dist, n = box_box([0, 0, 0], Ry, [h, h, h], [0, 0, 2 * e - 0.002], Rx, [h, h, h])
    assert len(pts) == 1 and dist[0] == pytest.approx(-0.002) and np.allclose(np.abs(n), [0, 0, 1]) and n[2] > 0
    assert np.allclose(pts[0], [0, 0, e - 0.001], atol=1e-12)
    pts, dist, _ = box_box([0, 0, 0], I, [0.5, 0.5, 0.1], [0.1, 0.05, 0.201], I, [0.1, 0.1, 0.1], margin=0.002)
    assert len(pts) == 4 and np.allclose(dist, 0.001)
    assert len(box_box([0, 0, 0], I, [0.5, 0.5, 0.1], [0.1, 0.05, 0.201], I, [0.1, 0.1, 0.1])[0]) == 0
    # tilted box: only the vertices that dip below the face (or within the margin) are contacts, and the depth is linear along the face
    t = 0.02
    Rt = np.array([[np.cos(t), 0, np.sin(t)], [0, 1, 0], [-np.sin(t), 0, np.cos(t)]])
    pts, dist, n = box_box([0, 0, 0], I, [0.5, 0.5, 0.1], [0, 0, 0.2], Rt, [0.1, 0.1, 0.1])
    assert len(pts) == 2 and np.allclose(n, [0, 0, 1])
    zc = 0.2 - 0.1 * np.cos(t) - 0.1 * np.sin(t) - 0.1                          # lowest corners: x = +h side dips
    assert np.allclose(dist, zc) and np.allclose(pts[:, 0], 0.1 * np.cos(t) - 0.1 * np.sin(t))
