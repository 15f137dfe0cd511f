/*
 * pmg_device.h -- device code of the per-step hot path, one wavefront per env.
 *
 * Replaces, for N envs at once, what the reference does through PyBullet in
 * Kuka.apply_action (P/robots/kuka.py:167-225): tip-target integration and
 * clipping, calculateInverseKinematics, POSITION_CONTROL motors and
 * 5 x stepSimulation (= 100 substeps of 2 ms with 5 solver iterations,
 * P/envs/base_envs/base_env.py:203-220), followed by the state read-out of
 * Kuka.calc_robot_state (kuka.py:227-256).
 *
 * Layout: lane l < 9 owns movable link / DoF l (iiwa_joint_1..7, finger1,
 * finger2) and keeps its link frame, spatial axis, inertia, velocity and its
 * row of M^-1 in registers.  Chain recursions are DPP prefix/suffix scans in
 * the first 16-lane row, matrix work is lane = row with v_readlane
 * broadcasts; no MFMA (nothing here is a dense contraction), no HBM traffic
 * inside the 100-substep loop.
 *
 * Spatial algebra: world coordinates, Pluecker vectors referenced to the
 * world origin, motion = [w; v_O], force = [n_O; f].  A rigid-body inertia is
 * the 10-vector (m, H = m c, Ibar about the origin: xx xy xz yy yz zz), so the
 * composite-body recursion is a plain suffix SUM along the chain.
 * Dynamics = CRBA (mass matrix) + RNEA (bias) + in-register Gauss-Jordan
 * M^-1, mathematically identical to the articulated-body algorithm the
 * oracle restates from Bullet.
 */
#ifndef PMG_DEVICE_H
#define PMG_DEVICE_H

#include <pmg_wave.h> /* -I csrc (gfx950) or -I tests/emu (CPU emulator) */
#include "../../include/pmg.h"
#include "../../include/pmg_model.h"

#define WV wv
namespace pmg {
#include "pmg_device_body.inc"
}  // namespace pmg
#undef WV
#define WV wr
namespace pmgp {
#include "pmg_device_body.inc"
}  // namespace pmgp
#undef WV
#endif
