"""Hand-built model blobs for closed-form known-answer tests of the oracle (no MJCF, no cosim_amd.compile involved).

Each builder fills a zeroed ``cosim_model_t`` field by field, the way MuJoCo's compiler would for the equivalent MJCF:
``body_invweight0`` / ``dof_invweight0`` of a single free body are 1/m and 1/I exactly, of a single hinge 1/(I_axis + armature).
"""
import ctypes

import numpy as np

from cosim_amd.model import CosimModel, set_field

MAGIC = 0x43534d31
DEFAULT_SOLREF = (0.02, 1.0)
DEFAULT_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)


class Tiny:
    """What ``oracle.Oracle`` reads from a compiled model: the blob plus (empty) side arrays."""

    def __init__(self, blob):
        self.blob = blob
        self.hull_vert = np.zeros((1, 3), np.float32)
        self.hull_adr = np.zeros(2, np.int32)
        self.hull_nbr = np.zeros(1, np.int32)
        self.hfield = np.zeros((2, 2), np.float32)


def _base(timestep=0.002, gravity=(0.0, 0.0, -9.81)):
    m = CosimModel()
    ctypes.memset(ctypes.addressof(m), 0, ctypes.sizeof(m))
    m.magic = MAGIC
    m.magic_end = MAGIC
    m.solver, m.iterations, m.ls_iterations, m.frame_skip = 0, 100, 50, 1
    m.timestep, m.tolerance, m.ls_tolerance, m.impratio = timestep, 1e-10, 0.01, 1.0
    set_field(m, "gravity", np.array(gravity, dtype=np.float64))
    m.ground_type, m.ground_contype, m.ground_conaffinity, m.ground_condim = 0, 1, 1, 3
    set_field(m, "ground_friction", np.array([1.0, 0.005, 0.0001]))
    set_field(m, "ground_solref", np.array(DEFAULT_SOLREF))
    set_field(m, "ground_solimp", np.array(DEFAULT_SOLIMP))
    m.ground_solmix = 1.0
    ident = np.zeros((32, 4)); ident[:, 0] = 1.0
    set_field(m, "body_quat", ident)
    set_field(m, "body_iquat", ident)
    set_field(m, "geom_quat", np.tile([1.0, 0, 0, 0], (40, 1)))
    set_field(m, "imu_quat", np.array([1.0, 0, 0, 0]))
    m.imu_bodyid = 1
    m.heightmap_miss = 1.0
    return m


def free_sphere(mass=2.0, radius=0.1, inertia=None, mu=1.0, ground_mu=None, solref=DEFAULT_SOLREF, solimp=DEFAULT_SOLIMP, timestep=0.002,
                gravity=(0.0, 0.0, -9.81), collide=True, connect_to_world=None, eq_solref=DEFAULT_SOLREF, eq_solimp=DEFAULT_SOLIMP):
    """One free body carrying a sphere geom above the ground plane z = 0; optionally a ``connect`` of its centre to a world point."""
    m = _base(timestep, gravity)
    I = (0.4 * mass * radius * radius) if inertia is None else inertia
    m.nq, m.nv, m.nu, m.nbody, m.njnt = 7, 6, 0, 2, 1
    m.ngeom = 1 if collide else 0
    set_field(m, "body_parentid", np.array([0, 0], np.int32)); set_field(m, "body_rootid", np.array([0, 1], np.int32))
    set_field(m, "body_jntnum", np.array([0, 1], np.int32)); set_field(m, "body_jntadr", np.array([-1, 0], np.int32))
    set_field(m, "body_dofnum", np.array([0, 6], np.int32)); set_field(m, "body_dofadr", np.array([-1, 0], np.int32))
    set_field(m, "body_mass", np.array([0.0, mass])); set_field(m, "body_inertia", np.array([[0, 0, 0], [I, I, I]], dtype=np.float64))
    set_field(m, "body_invweight0", np.array([[0, 0], [1.0 / mass, 1.0 / I]]))
    set_field(m, "jnt_type", np.array([0], np.int32)); set_field(m, "jnt_bodyid", np.array([1], np.int32))
    set_field(m, "jnt_axis", np.array([[0.0, 0, 1]]))
    q0 = np.zeros(7); q0[2] = radius; q0[3] = 1.0
    set_field(m, "qpos0", q0); set_field(m, "init_qpos", q0)
    set_field(m, "dof_bodyid", np.full(6, 1, np.int32)); set_field(m, "dof_jntid", np.zeros(6, np.int32))
    set_field(m, "dof_parentid", np.array([-1, 0, 1, 2, 3, 4], np.int32))
    set_field(m, "dof_invweight0", np.array([1 / mass] * 3 + [1 / I] * 3))
    set_field(m, "dof_solref", np.tile(DEFAULT_SOLREF, (6, 1))); set_field(m, "dof_solimp", np.tile(DEFAULT_SOLIMP, (6, 1)))
    m.meaninertia = (3 * mass + 3 * I) / 6.0
    if ground_mu is not None:
        set_field(m, "ground_friction", np.array([ground_mu, 0.005, 0.0001]))
    if collide:
        set_field(m, "geom_type", np.array([2], np.int32)); set_field(m, "geom_bodyid", np.array([1], np.int32))
        set_field(m, "geom_contype", np.array([1], np.int32)); set_field(m, "geom_conaffinity", np.array([1], np.int32))
        set_field(m, "geom_condim", np.array([3], np.int32)); set_field(m, "geom_ground", np.array([1], np.int32))
        set_field(m, "geom_size", np.array([[radius, 0, 0]])); set_field(m, "geom_friction", np.array([[mu, 0.005, 0.0001]]))
        set_field(m, "geom_solref", np.array([solref])); set_field(m, "geom_solimp", np.array([solimp]))
        set_field(m, "geom_solmix", np.array([1.0])); set_field(m, "geom_rbound", np.array([radius]))
        set_field(m, "geom_aabb", np.array([[0, 0, 0, radius, radius, radius]], dtype=np.float64))
        set_field(m, "ground_solref", np.array(solref)); set_field(m, "ground_solimp", np.array(solimp))
    if connect_to_world is not None:
        m.neq = 1
        set_field(m, "eq_body1", np.array([1], np.int32)); set_field(m, "eq_body2", np.array([0], np.int32))
        set_field(m, "eq_anchor1", np.zeros((1, 3))); set_field(m, "eq_anchor2", np.array([connect_to_world], dtype=np.float64))
        set_field(m, "eq_solref", np.array([eq_solref])); set_field(m, "eq_solimp", np.array([eq_solimp]))
    return Tiny(m)


def hinge_arm(mass=1.5, length=0.4, armature=0.01, damping=0.0, frictionloss=0.0, limited=False, jrange=(-0.5, 0.5), axis=(0.0, 1.0, 0.0),
              solref=DEFAULT_SOLREF, solimp=DEFAULT_SOLIMP, timestep=0.002, gravity=(0.0, 0.0, -9.81), motor=False):
    """One body on a hinge fixed in the world at the origin, point-like mass (small inertia) at distance ``length`` along +x."""
    m = _base(timestep, gravity)
    Ic = 1e-4
    m.nq, m.nv, m.nbody, m.njnt, m.ngeom = 1, 1, 2, 1, 0
    m.nu = 1 if motor else 0
    set_field(m, "body_parentid", np.array([0, 0], np.int32)); set_field(m, "body_rootid", np.array([0, 1], np.int32))
    set_field(m, "body_jntnum", np.array([0, 1], np.int32)); set_field(m, "body_jntadr", np.array([-1, 0], np.int32))
    set_field(m, "body_dofnum", np.array([0, 1], np.int32)); set_field(m, "body_dofadr", np.array([-1, 0], np.int32))
    set_field(m, "body_mass", np.array([0.0, mass])); set_field(m, "body_inertia", np.array([[0, 0, 0], [Ic, Ic, Ic]], dtype=np.float64))
    set_field(m, "body_ipos", np.array([[0, 0, 0], [length, 0, 0]], dtype=np.float64))
    Iax = Ic + mass * length * length + armature
    set_field(m, "body_invweight0", np.array([[0, 0], [1.0 / mass, 1.0 / Iax]]))
    set_field(m, "jnt_type", np.array([3], np.int32)); set_field(m, "jnt_bodyid", np.array([1], np.int32))
    set_field(m, "jnt_axis", np.array([axis], dtype=np.float64))
    set_field(m, "jnt_limited", np.array([int(limited)], np.int32)); set_field(m, "jnt_range", np.array([jrange], dtype=np.float64))
    set_field(m, "jnt_solref", np.array([solref])); set_field(m, "jnt_solimp", np.array([solimp]))
    set_field(m, "dof_bodyid", np.array([1], np.int32)); set_field(m, "dof_jntid", np.array([0], np.int32))
    set_field(m, "dof_parentid", np.array([-1], np.int32))
    set_field(m, "dof_armature", np.array([armature])); set_field(m, "dof_damping", np.array([damping]))
    set_field(m, "dof_frictionloss", np.array([frictionloss])); set_field(m, "dof_invweight0", np.array([1.0 / Iax]))
    set_field(m, "dof_solref", np.array([solref])); set_field(m, "dof_solimp", np.array([solimp]))
    m.meaninertia = Iax
    if motor:
        set_field(m, "act_jntid", np.array([0], np.int32)); set_field(m, "act_dofid", np.array([0], np.int32))
        set_field(m, "act_gear", np.array([1.0]))
    t = Tiny(m)
    t.inertia_axis = Iax
    return t
