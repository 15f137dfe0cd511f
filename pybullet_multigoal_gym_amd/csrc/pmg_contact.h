/*
 * pmg_contact.h -- contacts of the batched env: box-box narrowphase (one pair
 * per lane), contact/friction constraint rows staged in LDS, and the
 * sequential-impulse visit of one row executed by the whole wavefront
 * (lane = DoF: lanes 0..8 robot joints, lanes 9+6b+c component c of block b).
 *
 * Restates, per env, what Bullet does inside stepSimulation for the pairs
 * that can touch in the reference's tasks: table x
 * block, block x block, finger x block, finger x table; rows and solver order
 * as in the oracle (oracle/pmg_oracle.c "collide", "row_setup", "substep").
 */
#ifndef PMG_CONTACT_H
#define PMG_CONTACT_H

#include <type_traits>

#include "pmg_device.h"

#define WV wv
namespace pmg {
#include "pmg_contact_body.inc"
}  // namespace pmg
#undef WV
#define WV wr
namespace pmgp {
#include "pmg_contact_body.inc"
}  // namespace pmgp
#undef WV
#endif
