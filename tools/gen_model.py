#!/usr/bin/env python3
"""Generate include/pmg_model.h (model constants) and tests/golden/model.json.

Build-owned generator: reads the reference's *asset data* (URDF numbers, binary
STL vertex extents) from /root/reference IN THIS CONTAINER ONLY and emits the
constants the hot path needs.  The emitted header is committed; nothing here
runs on the GPU box.  Citations (P/ = /root/reference/pybullet_multigoal_gym/):

  P/assets/robots/kuka/iiwa14_parallel_jaw.urdf   joints :94-288, tip :311-315,
      gripper base :390-414, fingers :415-478, tabs :480-523
  P/assets/objects/table.urdf, block.urdf, long_table.urdf, cylinder_bulk.urdf

[BULLET-PRIOR] (pybullet~=3.0.6, absent from the container; restated from its
published behaviour): loadURDF without URDF_USE_INERTIA_FROM_FILE recomputes
each link's inertia from the AABB of its collision compound
(btCompoundShape::calculateLocalInertia: m/12*(ly^2+lz^2, ...)), multiplies by
<inertia_scaling>, and leaves mass-0 links with the URDF's inertia diagonal
(here 0.1 kg m^2 on the tip/mocap/cam/tab helper links).  Convex-hull children
carry the 0.001 URDF collision margin twice in their AABB (recalcLocalAabb +
getAabb); box/cylinder children are exact.
"""
import json, math, os, struct, sys
import xml.etree.ElementTree as ET
import numpy as np

REF = '/root/reference/pybullet_multigoal_gym/assets'
OUT_H = os.path.join(os.path.dirname(__file__), '..', 'include', 'pmg_model.h')
OUT_J = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'model.json')
MARGIN = 0.001


def vec(s):
    return [float(t.strip(',')) for t in s.replace(',', ' ').split()]


def rpy_to_R(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    R = np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                  [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                  [-sp, cp * sr, cp * cr]])
    Rc = np.round(R)  # every fixed rotation in this URDF is a multiple of pi/2
    assert np.abs(R - Rc).max() < 1e-9, R
    return Rc + 0.0


def stl_extent(path):
    b = open(path, 'rb').read()
    n = struct.unpack('<I', b[80:84])[0]
    assert len(b) == 84 + 50 * n
    a = np.frombuffer(b[84:], dtype=np.dtype([('n', '<f4', 3), ('v', '<f4', (3, 3)), ('a', '<u2')]))
    v = a['v'].reshape(-1, 3).astype(np.float64)
    return v.min(0), v.max(0)


def box_inertia(m, l):
    lx, ly, lz = l
    return [m / 12.0 * (ly * ly + lz * lz), m / 12.0 * (lx * lx + lz * lz), m / 12.0 * (lx * lx + ly * ly)]


def parse_robot():
    root = ET.parse(os.path.join(REF, 'robots/kuka/iiwa14_parallel_jaw.urdf')).getroot()
    links = {}
    for L in root.findall('link'):
        name = L.get('name')
        ine = L.find('inertial')
        mass = float(ine.find('mass').get('value'))
        org = ine.find('origin')
        com = vec(org.get('xyz')) if org is not None else [0, 0, 0]
        assert org is None or vec(org.get('rpy', '0 0 0')) == [0, 0, 0]
        I = ine.find('inertia')
        urdf_inertia = [float(I.get('ixx')), float(I.get('iyy')), float(I.get('izz'))]
        col = L.find('collision')
        ext = None
        kind = None
        aabb = None     # collision shape's bounding box in the link frame (tools/arm_link_penetration.py)
        if col is not None:
            corg = col.find('origin')
            assert corg is None or (vec(corg.get('xyz')) == [0, 0, 0] and vec(corg.get('rpy')) == [0, 0, 0])
            g = col.find('geometry')[0]
            if g.tag == 'mesh':
                lo, hi = stl_extent(os.path.join(REF, 'robots/kuka', g.get('filename')))
                ext = (hi - lo + 4 * MARGIN).tolist()
                kind = 'hull'
                aabb = [lo.tolist(), hi.tolist()]
            elif g.tag == 'box':
                ext = vec(g.get('size'))
                kind = 'box'
                aabb = [[-x / 2 for x in ext], [x / 2 for x in ext]]
            elif g.tag == 'cylinder':
                r, l = float(g.get('radius')), float(g.get('length'))
                ext = [2 * r, 2 * r, l]
                kind = 'cyl'
                aabb = [[-r, -r, -l / 2], [r, r, l / 2]]
        scaling = 1.0
        fric = None
        c = L.find('contact')
        if c is not None:
            if c.find('inertia_scaling') is not None:
                scaling = float(c.find('inertia_scaling').get('value'))
            if c.find('lateral_friction') is not None:
                fric = float(c.find('lateral_friction').get('value'))
        if mass > 0 and ext is not None:
            inertia = [scaling * x for x in box_inertia(mass, ext)]
        else:
            inertia = urdf_inertia  # mass-0 helper links keep the URDF diagonal
        links[name] = dict(mass=mass, com=com, inertia=inertia, ext=ext, kind=kind, fric=fric, aabb=aabb)
    joints = []
    for J in root.findall('joint'):
        o = J.find('origin')
        ax = J.find('axis')
        lim = J.find('limit')
        dyn = J.find('dynamics')
        joints.append(dict(name=J.get('name'), type=J.get('type'), parent=J.find('parent').get('link'),
                           child=J.find('child').get('link'), xyz=vec(o.get('xyz')), rpy=vec(o.get('rpy')),
                           axis=vec(ax.get('xyz')) if ax is not None else None,
                           lower=float(lim.get('lower')) if lim is not None else None,
                           upper=float(lim.get('upper')) if lim is not None else None,
                           damping=float(dyn.get('damping')) if dyn is not None else 0.0))
    return links, joints


def quicksort_all_equal(n):
    """Permutation produced by btAlignedObjectArray::quickSort when every key
    compares equal (all constraints of one multibody share an island id)."""
    a = list(range(n))

    def qs(lo, hi):
        i, j = lo, hi
        while True:
            # comparator is strict '<' on equal keys: never advances i or j
            if i <= j:
                a[i], a[j] = a[j], a[i]
                i += 1
                j -= 1
            if not (i <= j):
                break
        if lo < j:
            qs(lo, j)
        if i < hi:
            qs(i, hi)
    if n > 1:
        qs(0, n - 1)
    return a


def main():
    links, joints = parse_robot()
    movable = ['iiwa_joint_%d' % i for i in range(1, 8)] + ['iiwa_gripper_finger1_joint', 'iiwa_gripper_finger2_joint']
    jby = {j['name']: j for j in joints}
    child_of = {j['child']: j for j in joints}
    mov_link = [jby[n]['child'] for n in movable]

    # frame of every link relative to its nearest movable ancestor link
    def rel_to_movable(link):
        R = np.eye(3)
        t = np.zeros(3)
        cur = link
        while cur not in mov_link:
            j = child_of.get(cur)
            if j is None:
                return None, R, t  # attached to the static base
            Rj = rpy_to_R(*j['rpy'])
            t = Rj @ t + np.array(j['xyz'])
            R = Rj @ R
            cur = j['parent']
        return cur, R, t

    sub = {m: [] for m in mov_link}
    frames = {}
    for name, L in links.items():
        anc, R, t = rel_to_movable(name)
        frames[name] = (anc, R.tolist(), t.tolist())
        if anc is None:
            continue
        assert np.allclose(R, np.eye(3))
        com = (np.array(t) + R @ np.array(L['com'])).tolist()
        sub[anc].append(dict(name=name, mass=L['mass'], com=com, inertia=L['inertia']))

    # joint origin of movable joint k relative to its movable parent link
    jparent, jxyz, jR, jaxis, jtype, jlo, jhi, jdamp = [], [], [], [], [], [], [], []
    for n in movable:
        j = jby[n]
        anc, R, t = rel_to_movable(j['parent'])
        Rj = rpy_to_R(*j['rpy'])
        xyz = R @ np.array(j['xyz']) + t
        jparent.append(-1 if anc is None else mov_link.index(anc))
        jxyz.append(xyz.tolist())
        jR.append((R @ Rj).tolist())
        jaxis.append(j['axis'])
        jtype.append(0 if j['type'] == 'revolute' else 1)
        jlo.append(j['lower'])
        jhi.append(j['upper'])
        jdamp.append(j['damping'])

    nsub = max(len(v) for v in sub.values())
    model = dict(
        movable=movable, mov_link=mov_link, jparent=jparent, jxyz=jxyz, jR=jR, jaxis=jaxis, jtype=jtype,
        jlo=jlo, jhi=jhi, jdamp=jdamp, sub={k: v for k, v in sub.items()},
        tip_off=frames['iiwa_gripper_tip'][2], gbase_off=frames['iiwa_gripper_base_link'][2],
        tab1_off=frames['iiwa_gripper_finger1_finger_tab_link'][2],
        tab2_off=frames['iiwa_gripper_finger2_finger_tab_link'][2],
        finger_half=[x / 2 for x in links['iiwa_gripper_finger1']['ext']],
        gbase_radius=0.05, gbase_halflen=0.02,
        finger_friction=links['iiwa_gripper_finger1']['fric'],
        row_order=quicksort_all_equal(18),
    )
    # objects
    def obj(fn):
        r = ET.parse(os.path.join(REF, 'objects', fn)).getroot()
        L = r.find('link')
        m = float(L.find('inertial/mass').get('value'))
        g = L.find('collision/geometry')[0]
        c = L.find('contact')
        fr = float(c.find('lateral_friction').get('value'))
        sc = float(c.find('inertia_scaling').get('value')) if c.find('inertia_scaling') is not None else 1.0
        if g.tag == 'box':
            ext = vec(g.get('size'))
        else:
            ext = [2 * float(g.get('radius'))] * 2 + [float(g.get('length'))]
        I = [sc * x for x in box_inertia(m, ext)] if m > 0 else [0, 0, 0]
        return dict(mass=m, ext=ext, friction=fr, inertia=I, shape=g.tag)
    # the robot URDF's own base link (iiwa14_parallel_jaw.urdf:37-58): a 5 x 5 x 0.002 m collision box under link_0, friction 1 --
    # the floor an object knocked off the table lands on.  The robot base sits at the world origin (robot_bases.py:39-40).
    assert links['plane']['kind'] == 'box' and links['plane']['mass'] == 0
    model['plane'] = dict(mass=0.0, ext=links['plane']['ext'], friction=links['plane']['fric'], inertia=[0, 0, 0], shape='box')
    model['table'] = obj('table.urdf')
    model['long_table'] = obj('long_table.urdf')
    model['block'] = obj('block.urdf')
    model['puck'] = obj('cylinder_bulk.urdf')
    for c in ['blue', 'green', 'purple', 'red', 'yellow']:
        assert obj('block_%s.urdf' % c) == model['block']

    # ---- chests (P/assets/objects/chest_front_sliding_door.urdf: ChestPush, chest_up_sliding_door.urdf:
    # ChestPickAndPlace; P/robots/chest.py:5-23).  Static walls (mass 0 base + fixed mass-0 children), one prismatic
    # door link carrying the handle cylinder through a fixed joint, three mass-less key-point links.  Everything is
    # expressed relative to the chest base frame, which is never rotated (chest.py:33). ----
    def rpy_exact(r, p, y):
        cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
        return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                         [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                         [-sp, cp * sr, cp * cr]])

    def chest(fn):
        r = ET.parse(os.path.join(REF, 'objects', fn)).getroot()
        L = {l.get('name'): l for l in r.findall('link')}
        J = {j.find('child').get('link'): j for j in r.findall('joint')}
        def frame(name):   # pose of a link in the base frame at q = 0
            if name not in J:
                return np.zeros(3), np.eye(3)
            j = J[name]
            pp, pR = frame(j.find('parent').get('link'))
            o = j.find('origin')
            return pp + pR @ np.array(vec(o.get('xyz'))), pR @ rpy_exact(*vec(o.get('rpy')))
        walls, door, handle = [], None, None
        dj = J['chest_door']
        assert dj.get('type') == 'prismatic'
        for name, l in L.items():
            col = l.find('collision')
            if col is None:
                continue
            assert vec(col.find('origin').get('xyz')) == [0, 0, 0] and vec(col.find('origin').get('rpy')) == [0, 0, 0]
            g = col.find('geometry')[0]
            pos, R = frame(name)
            if name == 'chest_door_handle':
                handle = dict(c=pos, R=R, radius=float(g.get('radius')), halflen=float(g.get('length')) / 2,
                              friction=float(l.find('contact/lateral_friction').get('value')))
            elif name == 'chest_door':
                assert np.allclose(R, np.eye(3))
                door = dict(c=pos, half=[x / 2 for x in vec(g.get('size'))])
            else:
                assert np.allclose(R, np.eye(3)) and float(l.find('inertial/mass').get('value')) == 0
                walls.append(dict(c=pos.tolist(), half=[x / 2 for x in vec(g.get('size'))]))
        mass = sum(float(L[n].find('inertial/mass').get('value')) for n in ['chest_door', 'chest_door_handle'])
        kps = [frame('chest_door_%s_keypoint' % k)[0] - door['c'] for k in ['left', 'right', 'handle']]  # chest.py:34-38
        lim = dj.find('limit')
        return dict(walls=walls, door_c=door['c'].tolist(), door_half=door['half'], axis=vec(dj.find('axis').get('xyz')),
                    lower=float(lim.get('lower')), upper=float(lim.get('upper')), mass=mass,
                    handle_c=(handle['c'] - door['c']).tolist(), handle_R=handle['R'].tolist(),
                    handle_radius=handle['radius'], handle_halflen=handle['halflen'], handle_friction=handle['friction'],
                    keypoints=[k.tolist() for k in kps])
    model['chest'] = [chest('chest_front_sliding_door.urdf'), chest('chest_up_sliding_door.urdf')]
    for c_ in model['chest']:
        assert c_['lower'] == 0 and sorted(np.abs(c_['axis']).tolist()) == [0, 0, 1]


    # ---- the full multibody link list in PyBullet's depth-first order (base = 'plane'),
    # fixed joints kept as 0-dof links [BULLET-PRIOR: no URDF_MERGE_FIXED_LINKS] ----
    children = {}
    for j in joints:
        children.setdefault(j['parent'], []).append(j)
    bl = []
    def walk(link, parent_idx):
        for j in children.get(link, []):
            idx = len(bl)
            L = links[j['child']]
            dof = movable.index(j['name']) if j['name'] in movable else -1
            bl.append(dict(name=j['child'], joint=j['name'], parent=parent_idx,
                           type={'revolute': 0, 'prismatic': 1, 'fixed': 2}[j['type']],
                           xyz=j['xyz'], R=rpy_to_R(*j['rpy']).tolist(), axis=j['axis'] or [0.0, 0.0, 0.0],
                           mass=L['mass'], com=L['com'], inertia=L['inertia'], dof=dof, col_aabb=L['aabb']))
            walk(j['child'], idx)
    walk('plane', -1)
    names = [b['joint'] for b in bl]
    assert names.index('iiwa_gripper_tip_joint') == 8 and names.index('iiwa_gripper_finger1_joint') == 13 \
        and names.index('iiwa_gripper_finger2_joint') == 15, names   # P/test/multigoal/ik_test.py:44,85
    model['bullet_links'] = bl

    os.makedirs(os.path.dirname(OUT_J), exist_ok=True)
    with open(OUT_J, 'w') as f:
        json.dump(model, f, indent=1, sort_keys=True)

    def arr(x):
        if isinstance(x, (list, tuple)):
            return '{' + ', '.join(arr(v) for v in x) + '}'
        if isinstance(x, int):
            return str(x)
        return repr(float(x))

    H = []
    H.append('/* GENERATED by tools/gen_model.py from the reference URDF/STL asset numbers -- do not edit.\n'
             ' * Pure data shared by oracle/ and the HIP kernels (see tools/gen_model.py for sources\n'
             ' * and the [BULLET-PRIOR] inertia rules).  All lengths m, masses kg, SI throughout. */')
    H.append('#ifndef PMG_MODEL_H\n#define PMG_MODEL_H')
    H.append('#define PMG_NJ 9            /* movable joints: iiwa_joint_1..7, finger1, finger2 */')
    H.append('#define PMG_NARM 7')
    H.append('#define PMG_NSUB %d         /* max rigid sub-bodies carried by one movable link */' % nsub)
    H.append('#define PMG_JPARENT ' + arr(jparent))
    H.append('#define PMG_JTYPE ' + arr(jtype) + '   /* 0 revolute, 1 prismatic */')
    H.append('#define PMG_JXYZ ' + arr(jxyz) + '   /* joint origin in parent movable-link frame */')
    H.append('#define PMG_JROT ' + arr(jR) + '   /* joint frame rotation (parent <- child at q=0), exact 0/+-1 */')
    H.append('#define PMG_JAXIS ' + arr(jaxis))
    H.append('#define PMG_JLO ' + arr(jlo))
    H.append('#define PMG_JHI ' + arr(jhi))
    H.append('#define PMG_JDAMP ' + arr(jdamp))
    cnt, sm, sc, si = [], [], [], []
    for m in mov_link:
        s = sub[m]
        cnt.append(len(s))
        pad = nsub - len(s)
        sm.append([x['mass'] for x in s] + [0.0] * pad)
        sc.append([x['com'] for x in s] + [[0.0, 0.0, 0.0]] * pad)
        si.append([x['inertia'] for x in s] + [[0.0, 0.0, 0.0]] * pad)
    H.append('#define PMG_SUB_COUNT ' + arr(cnt))
    H.append('#define PMG_SUB_MASS ' + arr(sm))
    H.append('#define PMG_SUB_COM ' + arr(sc) + '   /* in the movable link frame */')
    H.append('#define PMG_SUB_INERTIA ' + arr(si) + '   /* principal, link-aligned, about the sub-body COM */')
    # ---- merged movable bodies (what the HIP kernels use): fixed children folded in ----
    mb_mass, mb_h, mb_ilo, mb_dsum, mb_sm, mb_sc = [], [], [], [], [], []
    for m in mov_link:
        M = 0.0; h = np.zeros(3); I = np.zeros((3, 3)); D = np.zeros(3); sm = []; scs = []
        for x in sub[m]:
            l = np.array(x['com']); ms = x['mass']
            M += ms; h += ms * l
            I += np.diag(x['inertia']) + ms * (l @ l * np.eye(3) - np.outer(l, l))
            D += np.array(x['inertia'])
            if ms > 0:
                sm.append(ms); scs.append(l.tolist())
        assert len(sm) <= 2
        while len(sm) < 2:
            sm.append(0.0); scs.append([0.0, 0.0, 0.0])
        mb_mass.append(M); mb_h.append(h.tolist())
        mb_ilo.append([I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]])
        mb_dsum.append(D.tolist()); mb_sm.append(sm); mb_sc.append(scs)
    H.append('#define PMG_MB_MASS ' + arr(mb_mass) + '   /* merged movable bodies */')
    H.append('#define PMG_MB_H ' + arr(mb_h) + '   /* first moment sum(m_s l_s), link frame */')
    H.append('#define PMG_MB_ILO ' + arr(mb_ilo) + '   /* inertia about the link origin, link axes: xx xy xz yy yz zz */')
    H.append('#define PMG_MB_DSUM ' + arr(mb_dsum) + '   /* sum of sub-body principal inertias (angular damping) */')
    H.append('#define PMG_MB_SUBM_MASS ' + arr(mb_sm) + '   /* massive sub-bodies (linear damping) */')
    H.append('#define PMG_MB_SUBM_COM ' + arr(mb_sc))
    H.append('#define PMG_TIP_OFF ' + arr(model['tip_off']) + '   /* iiwa_gripper_tip in link_7 frame */')
    H.append('#define PMG_GBASE_OFF ' + arr(model['gbase_off']))
    H.append('#define PMG_TAB1_OFF ' + arr(model['tab1_off']) + '   /* in finger1 frame */')
    H.append('#define PMG_TAB2_OFF ' + arr(model['tab2_off']))
    H.append('#define PMG_FINGER_HALF ' + arr(model['finger_half']))
    H.append('#define PMG_GBASE_RADIUS 0.05\n#define PMG_GBASE_HALFLEN 0.02')
    H.append('#define PMG_FINGER_FRICTION %r' % model['finger_friction'])
    H.append('#define PMG_ROW_ORDER ' + arr(model['row_order']) +
             '   /* btAlignedObjectArray::quickSort permutation of [limit j0..8, motor j0..8] */')
    H.append('#define PMG_BL_N %d   /* PyBullet link list (getNumJoints), fixed joints kept */' % len(bl))
    H.append('#define PMG_BL_PARENT ' + arr([b['parent'] for b in bl]))
    H.append('#define PMG_BL_TYPE ' + arr([b['type'] for b in bl]) + '   /* 0 revolute, 1 prismatic, 2 fixed */')
    H.append('#define PMG_BL_DOF ' + arr([b['dof'] for b in bl]))
    H.append('#define PMG_BL_XYZ ' + arr([b['xyz'] for b in bl]))
    H.append('#define PMG_BL_ROT ' + arr([b['R'] for b in bl]))
    H.append('#define PMG_BL_AXIS ' + arr([b['axis'] for b in bl]))
    H.append('#define PMG_BL_MASS ' + arr([b['mass'] for b in bl]))
    H.append('#define PMG_BL_COM ' + arr([b['com'] for b in bl]))
    H.append('#define PMG_BL_INERTIA ' + arr([b['inertia'] for b in bl]))
    H.append('#define PMG_BL_TIP 8\n#define PMG_BL_GBASE 12\n#define PMG_BL_FINGER1 13\n#define PMG_BL_TAB1 14\n#define PMG_BL_FINGER2 15\n#define PMG_BL_TAB2 16')
    for k in ['plane', 'table', 'long_table', 'block', 'puck']:
        o = model[k]
        K = k.upper()
        H.append('#define PMG_%s_HALF %s' % (K, arr([x / 2 for x in o['ext']])))
        H.append('#define PMG_%s_FRICTION %r' % (K, o['friction']))
        if o['mass'] > 0:
            H.append('#define PMG_%s_MASS %r' % (K, o['mass']))
            H.append('#define PMG_%s_INERTIA %s' % (K, arr(o['inertia'])))
    ch = model['chest']
    H.append('/* chests, index 0 = front sliding door (chest_push), 1 = up sliding lid (chest_pick_and_place); positions relative\n'
             ' * to the chest base (kuka_multi_step_base_env.py:64), walls padded to 4 with zero-size boxes */')
    H.append('#define PMG_CHEST_BASE {-0.7, 0.0, 0.21}')
    H.append('#define PMG_CHEST_NWALL ' + arr([len(c_['walls']) for c_ in ch]))
    pad = lambda w: w + [dict(c=[0.0, 0.0, -10.0], half=[0.0, 0.0, 0.0])] * (4 - len(w))
    H.append('#define PMG_CHEST_WALL_C ' + arr([[w['c'] for w in pad(c_['walls'])] for c_ in ch]))
    H.append('#define PMG_CHEST_WALL_HALF ' + arr([[w['half'] for w in pad(c_['walls'])] for c_ in ch]))
    H.append('#define PMG_CHEST_DOOR_C ' + arr([c_['door_c'] for c_ in ch]) + '   /* door box centre at q = 0 */')
    H.append('#define PMG_CHEST_DOOR_HALF ' + arr([c_['door_half'] for c_ in ch]))
    H.append('#define PMG_CHEST_DOOR_AXIS ' + arr([c_['axis'] for c_ in ch]))
    H.append('#define PMG_CHEST_DOOR_UPPER ' + arr([c_['upper'] for c_ in ch]))
    H.append('#define PMG_CHEST_DOOR_MASS ' + arr([c_['mass'] for c_ in ch]) + '   /* door + handle (fixed joint) */')
    H.append('#define PMG_CHEST_HANDLE_C ' + arr([c_['handle_c'] for c_ in ch]) + '   /* cylinder centre relative to the door centre */')
    H.append('#define PMG_CHEST_HANDLE_R ' + arr([c_['handle_R'] for c_ in ch]) + '   /* rpy (0, 1.57, 0): the axis is the third column */')
    H.append('#define PMG_CHEST_HANDLE_RADIUS ' + arr([c_['handle_radius'] for c_ in ch]))
    H.append('#define PMG_CHEST_HANDLE_HALFLEN ' + arr([c_['handle_halflen'] for c_ in ch]))
    H.append('#define PMG_CHEST_HANDLE_FRICTION %r' % ch[0]['handle_friction'])
    H.append('#define PMG_CHEST_WALL_FRICTION 0.5   /* no <contact> tag: [BULLET-PRIOR] btCollisionObject default */')
    H.append('#define PMG_CHEST_KEYPOINTS ' + arr([c_['keypoints'] for c_ in ch]) + '   /* left, right, handle (chest.py:34-38), relative to the door centre */')
    H.append('#endif')
    with open(OUT_H, 'w') as f:
        f.write('\n'.join(H) + '\n')
    print('wrote', OUT_H, OUT_J)
    for m in mov_link:
        print(m, [(s['name'], s['mass'], [round(x, 5) for x in s['inertia']]) for s in sub[m]])
    print('row order', model['row_order'])


if __name__ == '__main__':
    main()
