/* rkfd_device.h - the batched rkFDUpdate step, one world instance per 64-lane wavefront.
 *
 * This is the MI355X-native restatement of the reference hot path
 *   rkFDUpdate -> zODE2Update(RKG) -> _rkFDUpdate -> {FK, CD, contact solver, ABA}
 *   (reference src/rkfd_sim.c:525-566; src/rkfd_util.c; src/rkfd_penalty.c:11-31;
 *    src/rkfd_mlcp.c), designed for a wavefront rather than translated:
 *   - all spatial quantities are expressed in ONE world frame (Pluecker coordinates at
 *     the world origin, (angular, linear) ordering), so the three ABA sweeps need no
 *     6x6 congruence transforms: parents simply sum their children's articulated
 *     inertias;
 *   - forward kinematics and link velocities are log-depth pointer-jumping scans over
 *     lanes (lane = link) instead of serial recursions;
 *   - sweep 2 / sweep 3 run level-synchronously, 8 lanes per link (lane = row of the
 *     6x6), up to 8 links of a level at once, 6x6 data staged in LDS, 6-lane
 *     reductions done with DPP;
 *   - contact candidates, penalty forces, MLCP probe columns and the PGS residual
 *     update are lane-parallel (lane = candidate / contact / probe column / row).
 * The file compiles for gfx950 (hipcc) and, with -DRKFD_EMU, under a 64-thread lane
 * emulator that exists only so that tests can exercise the kernel logic without a GPU
 * (tests/emu; never part of the product library).
 */
#ifndef RKFD_DEVICE_H
#define RKFD_DEVICE_H

#ifndef __HIPCC_RTC__
#include <math.h>
#endif
#ifndef HUGE_VAL            /* (hipRTC has no system headers) */
#define HUGE_VAL __builtin_huge_val()
#endif
#include "rkfd_model.h"
#include "rkfd_devmodel.h"

#include "device/rkfd_dev_base.h"
#include "device/rkfd_dev_kinematics.h"
#include "device/rkfd_dev_sweeps.h"
#include "device/rkfd_dev_brf.h"
#include "device/rkfd_dev_contact.h"
#include "device/rkfd_dev_vertqp.h"
#include "device/rkfd_dev_mlcp.h"
#include "device/rkfd_dev_volume.h"
#include "device/rkfd_dev_step.h"

#endif /* RKFD_DEVICE_H */
