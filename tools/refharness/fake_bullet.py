"""A scripted Bullet client for the reference's Python (see the package docstring).

Only the ~25 PyBullet entry points the reference's hot path calls exist.  Names, indices,
limits and link kinematics come from the reference's URDF files, parsed here; the physics state lives in ONE world of
oracle/libpmg_oracle.so and is read / written through its pmgo_bw_* entry points.  Link states returned to the
reference are the oracle's; every one of them is cross-checked against an independent forward kinematics of the URDF
tree computed in this file (float64 numpy), so the oracle's model constants are re-validated against the URDF text at
every call.
"""
import ctypes as C
import math
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_lib  # noqa: E402  (the checker's ctypes binding; this whole package is test infrastructure)

CONSTANTS = dict(GUI=1, DIRECT=2, SHARED_MEMORY=3, POSITION_CONTROL=2, VELOCITY_CONTROL=0, TORQUE_CONTROL=1,
                 COV_ENABLE_GUI=1, ER_BULLET_HARDWARE_OPENGL=131072, URDF_USE_SELF_COLLISION=8,
                 JOINT_REVOLUTE=0, JOINT_PRISMATIC=1, JOINT_FIXED=4)
JOINT_TYPES = {'revolute': 0, 'prismatic': 1, 'fixed': 4}

# what the next BulletClient() simulates: set by the generator before it calls the reference's make_env
NEXT_WORLD = {}
LAST_CLIENT = None


def configure(**oracle_config):
    NEXT_WORLD.clear()
    NEXT_WORLD.update(oracle_config)


def _vec(s):
    return [float(t) for t in s.replace(',', ' ').split()]


def _rpy_R(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _axis_R(ax, q):
    x, y, z = ax
    c, s = math.cos(q), math.sin(q)
    t = 1 - c
    return np.array([[t * x * x + c, t * x * y - s * z, t * x * z + s * y],
                     [t * x * y + s * z, t * y * y + c, t * y * z - s * x],
                     [t * x * z - s * y, t * y * z + s * x, t * z * z + c]])


class UrdfTree:
    """Joints of one URDF in file order (= PyBullet's joint / link indices), with float64 forward kinematics."""

    def __init__(self, path):
        self.path = path
        root = ET.parse(path).getroot()
        self.name = root.get('name')
        self.link_com = {}
        for L in root.findall('link'):
            ine = L.find('inertial')
            org = ine.find('origin') if ine is not None else None
            self.link_com[L.get('name')] = np.array(_vec(org.get('xyz'))) if org is not None and org.get('xyz') else np.zeros(3)
        parsed = []
        children = set()
        for J in root.findall('joint'):
            o = J.find('origin')
            ax = J.find('axis')
            lim = J.find('limit')
            dyn = J.find('dynamics')
            d = dict(name=J.get('name'), type=J.get('type'), parent=J.find('parent').get('link'),
                     child=J.find('child').get('link'),
                     xyz=np.array(_vec(o.get('xyz'))) if o is not None and o.get('xyz') else np.zeros(3),
                     R=_rpy_R(*_vec(o.get('rpy'))) if o is not None and o.get('rpy') else np.eye(3),
                     axis=np.array(_vec(ax.get('xyz'))) if ax is not None else np.array([1.0, 0.0, 0.0]),
                     lower=float(lim.get('lower', 0)) if lim is not None else 0.0,
                     upper=float(lim.get('upper', -1)) if lim is not None else -1.0,
                     effort=float(lim.get('effort', 0)) if lim is not None else 0.0,
                     velocity=float(lim.get('velocity', 0)) if lim is not None else 0.0,
                     damping=float(dyn.get('damping', 0)) if dyn is not None else 0.0)
            assert d['type'] in JOINT_TYPES, d
            parsed.append(d)
            children.add(d['child'])
        links = [L.get('name') for L in root.findall('link')]
        bases = [n for n in links if n not in children]
        assert len(bases) == 1, bases
        self.base_link = bases[0]
        # [BULLET-PRIOR] URDF2Bullet numbers the links depth first (ComputeParentIndices), a link's child joints in
        # file order: that pre-order IS PyBullet's joint / link index
        self.joints = []
        child_index = {}

        def visit(link):
            for d in parsed:
                if d['parent'] == link:
                    d['index'] = len(self.joints)
                    child_index[d['child']] = d['index']
                    self.joints.append(d)
                    visit(d['child'])
        visit(self.base_link)
        assert len(self.joints) == len(parsed)
        dof = 0
        for d in self.joints:
            d['parent_index'] = child_index.get(d['parent'], -1)
            assert d['parent_index'] < d['index']
            d['dof'] = -1
            if d['type'] != 'fixed':
                d['dof'] = dof
                dof += 1
        self.num_dof = dof

    def fk(self, q, qd, base_pos):
        """World pose and velocity of every link's COM: (pos[3], R[3,3], lin[3], ang[3]) per joint index."""
        out, frames = [], []
        for d in self.joints:
            if d['parent_index'] >= 0:
                pp, pR, pv, pw = frames[d['parent_index']]
            else:
                pp, pR, pv, pw = np.asarray(base_pos, float), np.eye(3), np.zeros(3), np.zeros(3)
            p = pp + pR @ d['xyz']
            R0 = pR @ d['R']
            v = pv + np.cross(pw, p - pp)
            w = pw.copy()
            if d['type'] == 'revolute':
                aw = R0 @ d['axis']
                R = R0 @ _axis_R(d['axis'], q[d['dof']])
                w = w + aw * qd[d['dof']]
            elif d['type'] == 'prismatic':
                aw = R0 @ d['axis']
                p = p + aw * q[d['dof']]
                v = pv + np.cross(pw, p - pp) + aw * qd[d['dof']]
                R = R0
            else:
                R = R0
            frames.append((p, R, v, w))
            c = p + R @ self.link_com[d['child']]
            out.append((c, R, v + np.cross(w, c - p), w))
        return out


def _R_to_quat(m):
    """btMatrix3x3::getRotation (xyzw)."""
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    q = [0.0] * 4
    if tr > 0:
        s = math.sqrt(tr + 1.0)
        q[3] = 0.5 * s
        s = 0.5 / s
        q[0] = (m[2, 1] - m[1, 2]) * s
        q[1] = (m[0, 2] - m[2, 0]) * s
        q[2] = (m[1, 0] - m[0, 1]) * s
    else:
        i = 0 if m[0, 0] >= m[1, 1] else 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * s
        s = 0.5 / s
        q[3] = (m[k, j] - m[j, k]) * s
        q[j] = (m[j, i] + m[i, j]) * s
        q[k] = (m[k, i] + m[i, k]) * s
    return q


class _Body:
    def __init__(self, kind, path, pos, orn, tree=None, index=-1):
        self.kind, self.path, self.tree, self.index = kind, path, tree, index
        self.pos, self.orn = [float(x) for x in pos], [float(x) for x in orn]


class FakeBulletClient:
    """pybullet_utils.bullet_client.BulletClient, scripted (DIRECT mode only)."""

    def __init__(self, connection_mode=None):
        global LAST_CLIENT
        assert connection_mode in (None, CONSTANTS['DIRECT']), 'GUI rendering is outside the hot path'
        assert NEXT_WORLD, 'refharness.fake_bullet.configure(...) must name the world before the reference builds its env'
        self.__dict__.update(CONSTANTS)
        self._client = 0
        self.cfg_kw = dict(NEXT_WORLD)
        NEXT_WORLD.clear()
        self.task = self.cfg_kw['task']
        self.lib = oracle_lib.load(False)
        self.cfg = oracle_lib.make_config(num_envs=1, seed_base=0, seed_stride=0, **self.cfg_kw)
        self.h = C.c_void_p()
        rc = self.lib.pmgo_create(C.byref(self.cfg), C.byref(self.h))
        assert rc == 0, self.lib.pmgo_last_error(None)
        self.bodies = []
        self.nblocks = 0
        self.calls = {}
        self.params = {}
        self.fk_checks = 0
        self.fk_max_err = 0.0
        LAST_CLIENT = self

    # ------------------------------------------------------------------ bookkeeping
    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    def _bw(self, fn, *args):
        rc = getattr(self.lib, fn)(self.h, *args)
        assert rc >= 0, '%s failed (%d)' % (fn, rc)
        return rc

    def disconnect(self):
        if self.h:
            self.lib.pmgo_destroy(self.h)
            self.h = None

    # ------------------------------------------------------------------ world parameters (base_env.py:203-220)
    def setGravity(self, x, y, z):
        assert (x, y, z) == (0, 0, -9.81), (x, y, z)
        self.params['gravity'] = (x, y, z)

    def setDefaultContactERP(self, erp):
        assert erp == 0.9
        self.params['contact_erp'] = erp

    def setPhysicsEngineParameter(self, fixedTimeStep=None, numSolverIterations=None, numSubSteps=None, **kw):
        assert not kw, kw
        assert abs(fixedTimeStep - 0.04) < 1e-15 and numSolverIterations == 5 and numSubSteps == 20
        self.params.update(fixedTimeStep=fixedTimeStep, numSolverIterations=numSolverIterations, numSubSteps=numSubSteps)

    def setRealTimeSimulation(self, flag):
        assert not flag

    def computeViewMatrix(self, **kw):
        return tuple([0.0] * 16)

    def computeProjectionMatrixFOV(self, **kw):
        return tuple([0.0] * 16)

    def enableJointForceTorqueSensor(self, bodyUniqueId=None, jointIndex=None, enableSensor=None):
        assert not enableSensor

    # ------------------------------------------------------------------ loading
    def loadURDF(self, fileName, basePosition=(0, 0, 0), baseOrientation=(0, 0, 0, 1), useFixedBase=False, flags=0):
        self._count('loadURDF')
        name = os.path.basename(fileName)
        kind, tree, index = 'static', None, -1
        if name == 'iiwa14_parallel_jaw.urdf':
            kind, tree = 'robot', UrdfTree(fileName)
            assert tree.num_dof == 9 and len(tree.joints) == 17
        elif name.startswith('chest_'):
            want = {'chest_push': 'chest_front_sliding_door.urdf', 'chest_pick_and_place': 'chest_up_sliding_door.urdf'}
            assert want.get(self.task) == name, (self.task, name)
            kind, tree = 'chest', UrdfTree(fileName)
            assert tree.num_dof == 1
        elif name in ('block.urdf', 'cylinder_bulk.urdf') or name.startswith('block_'):
            assert (name == 'cylinder_bulk.urdf') == (self.task == 'slide')
            kind, index = 'block', self.nblocks
            self.nblocks += 1
        elif name in ('table.urdf', 'long_table.urdf'):
            assert (name == 'long_table.urdf') == (self.task == 'slide')
            kind = 'table'
        else:
            assert name.startswith('target'), name
        self.bodies.append(_Body(kind, fileName, basePosition, baseOrientation, tree, index))
        uid = len(self.bodies) - 1
        if kind == 'block':
            self.resetBasePositionAndOrientation(uid, basePosition, baseOrientation)
        return uid

    def getNumJoints(self, body):
        t = self.bodies[body].tree
        return len(t.joints) if t is not None else 0

    def getJointInfo(self, body, j):
        d = self.bodies[body].tree.joints[j]
        return (j, d['name'].encode(), JOINT_TYPES[d['type']], -1 if d['dof'] < 0 else 7 + d['dof'], -1 if d['dof'] < 0 else 6 + d['dof'],
                0, d['damping'], 0.0, d['lower'], d['upper'], d['effort'], d['velocity'], d['child'].encode(),
                tuple(d['axis']), tuple(d['xyz']), (0.0, 0.0, 0.0, 1.0), d['parent_index'])

    # ------------------------------------------------------------------ joint / link state
    def _dof(self, body, j):
        b = self.bodies[body]
        d = b.tree.joints[j]
        assert d['dof'] >= 0, 'joint %s is fixed' % d['name']
        return (0 if b.kind == 'robot' else 1), d['dof']

    def resetJointState(self, bodyUniqueId, jointIndex, targetValue, targetVelocity=0.0):
        self._count('resetJointState')
        bid, dof = self._dof(bodyUniqueId, jointIndex)
        self._bw('pmgo_bw_reset_joint', bid, dof, C.c_double(float(targetValue)), C.c_double(float(targetVelocity)))

    def getJointState(self, bodyUniqueId, jointIndex):
        self._count('getJointState')
        bid, dof = self._dof(bodyUniqueId, jointIndex)
        out = (C.c_double * 2)()
        self._bw('pmgo_bw_joint_state', bid, dof, out)
        return out[0], out[1], (0.0,) * 6, 0.0

    def _motor(self, body, j, mode, target, tvel, force, kp, kd):
        assert mode == CONSTANTS['POSITION_CONTROL'], 'only POSITION_CONTROL is used on the hot path'
        assert float(tvel) == 0.0
        bid, dof = self._dof(body, j)
        if float(force) != 0.0:
            # the restated motor has the gains of kuka.py:287-301 / chest.py:59-68 built in
            assert abs(float(kp) - 0.03) < 1e-15 and float(kd) == 1.0, (kp, kd)
        self._bw('pmgo_bw_motor', bid, dof, C.c_double(float(target)), C.c_double(float(force)))

    def setJointMotorControl2(self, bodyUniqueId=None, jointIndex=None, controlMode=None, targetPosition=0.0, targetVelocity=0.0,
                              force=None, positionGain=0.1, velocityGain=1.0, bodyIndex=None):
        self._count('setJointMotorControl2')
        body = bodyUniqueId if bodyUniqueId is not None else bodyIndex
        assert force is not None, 'the reference always passes force on the hot path'
        self._motor(body, jointIndex, controlMode, targetPosition, targetVelocity, force, positionGain, velocityGain)

    def setJointMotorControlArray(self, bodyUniqueId, jointIndices, controlMode, targetPositions=None, targetVelocities=None,
                                  forces=None, positionGains=None, velocityGains=None):
        self._count('setJointMotorControlArray')
        n = len(jointIndices)
        for a in (targetPositions, targetVelocities, forces, positionGains, velocityGains):
            assert len(a) == n
        for i in range(n):
            self._motor(bodyUniqueId, jointIndices[i], controlMode, targetPositions[i], targetVelocities[i], forces[i],
                        positionGains[i], velocityGains[i])

    def _robot_state(self):
        q, qd = np.zeros(9), np.zeros(9)
        out = (C.c_double * 2)()
        for d in range(9):
            self._bw('pmgo_bw_joint_state', 0, d, out)
            q[d], qd[d] = out[0], out[1]
        return q, qd

    def getLinkState(self, bodyUniqueId, linkIndex, computeLinkVelocity=0, computeForwardKinematics=0):
        self._count('getLinkState')
        b = self.bodies[bodyUniqueId]
        if b.kind == 'robot':
            out = (C.c_double * 13)()
            self._bw('pmgo_bw_link_state', int(linkIndex), out)
            o = np.array(out[:])
            q, qd = self._robot_state()
            c, R, v, w = b.tree.fk(q, qd, b.pos)[linkIndex]     # independent URDF kinematics: must agree
            err = max(np.abs(o[:3] - c).max(), np.abs(o[7:10] - v).max(), np.abs(o[10:13] - w).max(),
                      min(np.abs(o[3:7] - _R_to_quat(R)).max(), np.abs(o[3:7] + np.array(_R_to_quat(R))).max()))
            self.fk_checks += 1
            self.fk_max_err = max(self.fk_max_err, float(err))
            assert err < 1e-9, 'oracle link %d disagrees with the URDF kinematics by %g' % (linkIndex, err)
            pos, orn, lin, ang = tuple(o[:3]), tuple(o[3:7]), tuple(o[7:10]), tuple(o[10:13])
        else:
            assert b.kind == 'chest'
            jq = (C.c_double * 2)()
            self._bw('pmgo_bw_joint_state', 1, 0, jq)
            c, R, v, w = b.tree.fk([jq[0]], [jq[1]], b.pos)[linkIndex]
            pos, orn, lin, ang = tuple(c), tuple(_R_to_quat(R)), tuple(v), tuple(w)
        base = (pos, orn, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0), pos, orn)
        return base + (lin, ang) if computeLinkVelocity else base

    # ------------------------------------------------------------------ bases
    def resetBasePositionAndOrientation(self, body, pos, orn):
        self._count('resetBasePositionAndOrientation')
        b = self.bodies[body]
        assert b.kind in ('block', 'static'), 'the hot path never moves the %s' % b.kind
        b.pos, b.orn = [float(x) for x in pos], [float(x) for x in orn]
        if b.kind == 'block':
            self._bw('pmgo_bw_set_block', b.index, (C.c_double * 3)(*b.pos), (C.c_double * 4)(*b.orn))

    def _block(self, b):
        out = (C.c_double * 13)()
        self._bw('pmgo_bw_block_state', b.index, out)
        return out[:]

    def getBasePositionAndOrientation(self, body):
        self._count('getBasePositionAndOrientation')
        b = self.bodies[body]
        if b.kind == 'block':
            s = self._block(b)
            return tuple(s[:3]), tuple(s[3:7])
        return tuple(b.pos), tuple(b.orn)

    def getBaseVelocity(self, body):
        self._count('getBaseVelocity')
        b = self.bodies[body]
        assert b.kind == 'block'
        s = self._block(b)
        return tuple(s[7:10]), tuple(s[10:13])

    def getEulerFromQuaternion(self, q):
        x, y, z, w = q
        sarg = -2.0 * (x * z - w * y)
        if sarg <= -0.99999:
            return (0.0, -0.5 * math.pi, 2 * math.atan2(x, -y))
        if sarg >= 0.99999:
            return (0.0, 0.5 * math.pi, 2 * math.atan2(-x, y))
        return (math.atan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z), math.asin(sarg),
                math.atan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z))

    # ------------------------------------------------------------------ physics
    def calculateInverseKinematics(self, bodyUniqueId, endEffectorLinkIndex, targetPosition, targetOrientation=None,
                                   lowerLimits=None, upperLimits=None, jointRanges=None, restPoses=None,
                                   maxNumIterations=20, residualThreshold=1e-4):
        self._count('calculateInverseKinematics')
        b = self.bodies[bodyUniqueId]
        assert b.kind == 'robot' and b.tree.joints[endEffectorLinkIndex]['child'] == 'iiwa_gripper_tip'
        # the 7-entry null-space lists do not match the 9 DoFs: pybullet.c drops them and runs plain DLS [BULLET-PRIOR]
        assert len(lowerLimits) == len(upperLimits) == len(jointRanges) == len(restPoses) == 7
        out = (C.c_double * 9)()
        self._bw('pmgo_bw_ik', (C.c_double * 3)(*[float(x) for x in targetPosition]),
                 (C.c_double * 4)(*[float(x) for x in targetOrientation]), int(maxNumIterations),
                 C.c_double(float(residualThreshold)), out)
        return tuple(out[:])

    def stepSimulation(self):
        self._count('stepSimulation')
        self._bw('pmgo_bw_step_simulation')

    # ------------------------------------------------------------------ read-out for the generator
    def world_state(self):
        q, qd = self._robot_state()
        st = dict(q=q.tolist(), qd=qd.tolist())
        blocks = sorted((b for b in self.bodies if b.kind == 'block'), key=lambda b: b.index)
        st['blocks'] = [self._block(b) for b in blocks]
        if self.task.startswith('chest'):
            jq = (C.c_double * 2)()
            self._bw('pmgo_bw_joint_state', 1, 0, jq)
            st['door'] = [jq[0], jq[1]]
        return st
