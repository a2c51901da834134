/* rkfd_hip.h - C ABI of the MI355X (gfx950) batched rkFDUpdate path.
 *
 * Plain C: pointers, sizes and an opaque handle only (no torch / C++ types), so the
 * reference's C host code - or any FFI - can bind it.  Each entry point names the
 * reference interface it stands in for.  One process drives one GPU; a batch holds B
 * independent copies ("instances") of one world (all chains registered in one rkFD).
 *
 * Layout of every per-instance array handed across this boundary: instance-major,
 * x[b*stride + j] (b = instance).  Host pointers unless the name says "Dev".
 *
 * Every function returns 0 on success and a negative value on failure;
 * rkfdHipLastError() then describes the failure.  Without a usable GPU / kernel image
 * the calls FAIL (there is no CPU fallback).
 */
#ifndef RKFD_HIP_H
#define RKFD_HIP_H

#include "rkfd_model.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rkfdBatch rkfdBatch;

/* number of visible HIP devices (0 when none) */
int rkfdHipDeviceCount(void);
const char *rkfdHipLastError(void);

/* Creates device state for `batch` instances of world `m` on `device`.
 * Replaces, for the whole batch: rkFDCreate + rkFDChainReg* + rkFDUpdateInit's allocations
 * (reference src/rkfd_sim.c:32-54,188-235,552-558; rkFDCDUpdateInit src/rkfd_cd.c:22-31;
 * plugin _init src/rkfd_mlcp.c:312-325, src/rkfd_vert.c:350-368).  max_rigid = capacity of rigid
 * contact vertices solved per instance (MLCP plugin: 3*max_rigid <= 128; Vert plugin: pyramid*max_rigid
 * <= 192 - up to 8 vertices the QP keeps its factor in registers, up to 64 faces and unknowns one per lane, beyond that the
 * wide form with everything in LDS, one instance per CU; Volume plugin: rigid PAIRS in collision at once, at most 10 -
 * rigid pairs of convex polyhedra with at most 64 faces together are solved, the others are guarded:
 * status 4 when one comes into contact); exceeding it at run time is reported as an error by
 * rkfdBatchStatus. */
rkfdBatch *rkfdBatchCreate(const rkfdModel *m, int batch, int device, int max_rigid);
void rkfdBatchDestroy(rkfdBatch *b);

int rkfdBatchSize(const rkfdBatch *b);
int rkfdBatchDof(const rkfdBatch *b);

/* rkFDChainSetDis / rkFDChainSetVel (reference src/rkfd_sim.c:277-287), all instances at once:
 * dis, vel are [batch][ndof] */
int rkfdBatchSetState(rkfdBatch *b, const double *dis, const double *vel);
/* fd->dis, fd->vel, fd->acc after rkFDUpdate (reference include/roki_fd/rkfd_sim.h:49-50); any may be NULL */
int rkfdBatchGetState(rkfdBatch *b, double *dis, double *vel, double *acc);
/* rkJointMotorSetInput on every link (reference example/chain/arm_box_test.c:21): [batch][nlink] */
int rkfdBatchSetMotorInput(rkfdBatch *b, const double *input);
/* contact-vertex state rkCDVert{type,_ref} and force f per candidate vertex
 * (consumed at reference src/rkfd_util.c:256-263, src/rkfd_mlcp.c:263-279): [batch][ncand], [batch][ncand*3] */
/* (entries of a candidate that is not in contact are reported as 0; the anchors `ref` are in the frame of the
 * other cell's link as registered - links rigidly attached to it are merged on the device, the boundary converts) */
int rkfdBatchGetContact(rkfdBatch *b, int *active, int *type, double *ref, double *f);
int rkfdBatchSetContact(rkfdBatch *b, const int *active, const int *type, const double *ref);
/* joint friction pivots rkJointFrictionPivot{type,prev_trq} (reference src/rkfd_util.c:289-311): [batch][nlink] */
int rkfdBatchGetPivot(rkfdBatch *b, int *type, double *prev_trq);
int rkfdBatchSetPivot(rkfdBatch *b, const int *type, const double *prev_trq);

/* breakable float joints (reference example/model/wall.ztk:51-53; RoKi rk_joint_brfloat, include/rkfd_model.h): 1 per link whose
 * joint has broken, 0 elsewhere, [batch][nlink].  State like the friction pivots: every launch reads it and writes it back. */
int rkfdBatchGetBroken(rkfdBatch *b, int *broken);
int rkfdBatchSetBroken(rkfdBatch *b, const int *broken);

/* rkFDUpdateInit's committing evaluation _rkFDUpdateRef (reference src/rkfd_sim.c:542-549,556).
 * stream: hipStream_t (NULL = default stream).  Asynchronous. */
int rkfdBatchUpdateInit(rkfdBatch *b, void *stream);
/* nsteps x rkFDUpdate (reference src/rkfd_sim.c:560-566) for every instance: one launch with the steps fused, or -
 * under split launches - nsteps rounds of one-step launches on the internal streams (same results).  Asynchronous. */
int rkfdBatchUpdate(rkfdBatch *b, int nsteps, void *stream);
/* one evaluation _rkFDUpdate (doUpRef=0) or _rkFDUpdateRef (doUpRef=1) at the current state
 * (reference src/rkfd_sim.c:533-549): fills acc and the contact forces.  Asynchronous. */
int rkfdBatchEval(rkfdBatch *b, int doUpRef, void *stream);
/* Split launches: the batch goes out as `nsplit` (1..8) kernels over contiguous parts on internal streams.  The
 * instances are independent, so the parts need not wait for each other: the thinly occupied tail of one
 * step of one part then overlaps the next step of another (+20 % at 4096 instances per GPU).  The parts start
 * after whatever `stream` holds at the time of the call; `stream` itself does NOT wait for them until
 * rkfdBatchJoin( b, stream ) or rkfdBatchStatus( b, stream ); the host-side accessors (Get / Set) wait.
 * nsplit = 1 (default): one launch on the caller's stream, plain stream order. */
int rkfdBatchSetSplit(rkfdBatch *b, int nsplit);
/* Under split launches a call of n steps goes out as rounds of launches of at most `steps` steps each (default 5; worlds under
 * the Vert plugin keep one fused launch per part): one step per launch reloads the instance's state and the world's tables every
 * step, all n in one launch hold the slots while the rest of the batch waits - measured on config 4, rollouts of 25 steps, 1 / 2 /
 * 3 / 5 / 9 / 13 / 25 steps per launch: 11.51 / 11.63 / 11.65 / 11.82 / 11.79 / 11.82 / 11.76 M steps/s.  Results do not depend
 * on it (tests/test_gpu_edge.py: fused vs single-step launches bit for bit). */
int rkfdBatchSetStepsPerLaunch(rkfdBatch *b, int steps);
/* Compile the step kernel for THIS world (hipRTC, a few seconds): the same device code with the world's dimensions
 * as literals, so that the LDS layout, loop bounds and table strides fold into immediates (+7 % steps/s on the
 * 30-DoF humanoid).  Results are bit-identical to the generic kernels, which stay in use for the profiling entry
 * point.  Self-contained: the library carries the device sources it hands to hipRTC, and binds hipRTC and the compiler
 * library beside it (the ROCm it was built with; RKFD_ROCM_LIBDIR overrides) in a private link namespace, so neither files
 * beside the library nor the order in which the host program loaded other ROCm-bundling libraries matter; 0, or -1 with
 * a message (the generic kernels remain in use then).  rkfdSpecializeCompile: the compile step alone, without a GPU - bytes
 * of code object, or -1. */
int rkfdBatchSpecialize(rkfdBatch *b);
int rkfdSpecializeCompile(const rkfdModel *m, int max_rigid);
int rkfdSpecializeCompileW(const rkfdModel *m, int max_rigid, int ipw);      /* ipw: instances per wavefront, 1 or 2 (below) */
/* Ahead-of-time kernels: a specialised kernel is a function of the world's dimensions and the device sources only, so its code
 * object is kept in a directory beside the library (`spec/`; RKFD_SPEC_DIR overrides, RKFD_SPEC_STORE=0 switches the store off),
 * keyed by a hash of everything that goes into it.  `make spec` fills it for the worlds of BASELINE.json's configurations at
 * build time, so that those need no run-time compiler at all; other worlds are compiled on first use and kept.  1 when the
 * last rkfdBatchSpecialize / rkfdSpecializeCompile was served from the store, else 0 (measurement / test aid). */
int rkfdSpecializeLastFromStore(void);
/* test aid: XOR mask over the kernel variants of the batches created AFTER the call (4: the Vert QP's Q = A'A off the matrix
 * cores, 8: grouped Gauss-Seidel off, 32: its sweep-order storage off); returns the previous mask.  0 = the product's defaults. */
int rkfdDebugVariants(int mask);
/* Instances per wavefront of the world-specific kernel: 1 (default: one instance has the 64 lanes) or 2 (two instances share a
 * wavefront, 32 lanes each: half the instruction issue per instance where an instance leaves most of the 64 lanes idle).  Call
 * before rkfdBatchSpecialize; needs a world of at most 32 device links, 32 joint coordinates, 32 contact slots and 16 rigid
 * contact vertices under the MLCP plugin or with elastic contacts only (else -1 with a message, one per wavefront stays).
 * Results are bit-identical to one instance per wavefront.  rkfdBatchInstancesPerWave: what the launches really use. */
int rkfdBatchSetInstancesPerWave(rkfdBatch *b, int ipw);
int rkfdBatchInstancesPerWave(const rkfdBatch *b);
/* MEASURE which of the two is faster for this world, on this device, at this batch size, and keep it: nsteps steps under
 * either from the batch's present state (set one and call rkfdBatchUpdateInit first), which is put back afterwards; an
 * earlier rkfdBatchSnapshot stays as it was.  Returns the chosen count, -1 on error; ms[2] (may be NULL): the milliseconds
 * measured for 1 and 2 (ms[1] < 0: the world is not eligible for two).  Safe by construction: both give the same bits. */
int rkfdBatchTuneInstancesPerWave(rkfdBatch *b, int nsteps, double *ms);
int rkfdBatchJoin(rkfdBatch *b, void *stream);
/* measurement aid: with on = 1 every launch is bracketed by HIP events on the stream it runs on;
 * rkfdBatchLaunchTiming synchronises the device and returns their number and summed duration */
int rkfdBatchTimeLaunches(rkfdBatch *b, int on);
int rkfdBatchLaunchTiming(rkfdBatch *b, int *launches, double *total_ms);

/* waits for the stream-ordered work, then reports device-side conditions:
 * 0 ok, 1 rigid contact met without a rigid solver set up (max_rigid = 0),
 * 2 contact capacity exceeded - more rigid contact vertices (Volume plugin: pairs in collision, or contact-plane
 * conditions of a pair: 8) than max_rigid, or more rigid + elastic contact
 * vertices than the active-contact slots (max_rigid when the world has no elastic pairs, else max(max_rigid, 16)
 * capped by the candidate count); the vertices beyond the capacity were dropped -, 3 the Vert plugin's QP ran
 * out of iterations (256) or of basis history (64), 4 Volume plugin: a GUARDED pair came into contact - a rigid pair with a
 * shape that is not convex, or with more than 64 faces together, cannot be clipped on the device; such a world is accepted
 * (the reference's humanoid mighty.ztk, whose body meshes are not convex, stands and walks on its convex soles) and the
 * plugin's own collision test watches those pairs -; negative: HIP error.  rkfdHipLastError() describes a
 * non-zero status.  A condition is reported ONCE: the call clears the device-side flag, so the next status tells
 * what happened after this one (when several conditions occurred since the last call, the last one written wins). */
int rkfdBatchStatus(rkfdBatch *b, void *stream);

/* Mean number of rigid / elastic contact vertices per instance at the committing evaluation of the rkFDUpdate
 * steps run since the last reset (the length of the plugin's vertex lists, reference src/rkfd_mlcp.c:327-345,
 * src/rkfd_vert.c:380-392), and the number of instance-steps counted.  Measurement aid: tells what contact
 * problem a timed region really solved.  Synchronises the device.  Any output may be NULL. */
int rkfdBatchContactStats(rkfdBatch *b, int reset, double *mean_rigid, double *mean_elastic, long long *instance_steps);

/* MPC-style rollouts: rkfdBatchSnapshot keeps a device-resident copy of the whole per-instance state (joint
 * state, accelerations, friction pivots, contact-vertex state and forces; synchronous), rkfdBatchRestore puts it
 * back in stream order (one small copy kernel per part, no host traffic) - the start of the next rollout from the
 * same states.  The reference has no counterpart: a caller would re-run rkFDChainSetDis / SetVel per instance
 * (reference src/rkfd_sim.c:277-287) and rkFDUpdateInit. */
int rkfdBatchSnapshot(rkfdBatch *b);
int rkfdBatchRestore(rkfdBatch *b, void *stream);

/* diagnostic launch: nsteps x rkFDUpdate with in-kernel phase stamps.  out is [batch][32]
  * (RKFD_NPROF = 32 per instance) shader-clock cycles: kinematics, collision+penalty, sweep 2, sweep 3 (both
 * passes), MLCP, tail, MLCP matrix, whole launch, then finer stamps inside sweep 2 (8-13), MLCP (14, 15, 21-23),
 * kinematics (16-20) and the Vert QP (24-30); tools/prof_phases.py names them.
 * Synchronous; not for timing runs (the stamps serialise the phases). */
int rkfdBatchProfile(rkfdBatch *b, int nsteps, unsigned long long *out);

/* device pointers to the live state ([batch][ndof] doubles), for zero-copy consumers
 * (e.g. an RCCL all-gather of final states) */
double *rkfdBatchDevDis(rkfdBatch *b);
double *rkfdBatchDevVel(rkfdBatch *b);
double *rkfdBatchDevAcc(rkfdBatch *b);

/* kernel resource facts for measurement: LDS bytes per instance */
int rkfdBatchLdsBytes(const rkfdBatch *b);
/* instances that can be resident on one compute unit at a time: the HIP runtime's occupancy answer for the step
 * kernel (registers, LDS), corrected for the 1280-byte pieces in which the hardware allocates LDS; -1 on error */
int rkfdBatchResidency(const rkfdBatch *b);
/* the same figure computed on the host for a model and contact capacity (no GPU needed) */
int rkfdLdsBytesFor(const rkfdModel *m, int max_rigid);

/* ---- the node level: all the GPUs of one node from ONE process, host code in C -------------------------------------
 * SURVEY 8e / north star: "independent MPC-style rollouts shard embarrassingly across the 8 GPUs of one node with an
 * RCCL-over-xGMI gather of final states only".  The instances of a batch are independent (one rkFD never references another),
 * so device k simulates the contiguous block [lo, hi) of the `total` instances (remainders to the low devices), model constants
 * replicated, NO per-step communication; every device has its own host thread and HIP stream inside the library, and the only
 * collective is one ncclAllGather of the final {dis, vel} per rollout (librccl is bound at run time; RKFD_RCCL_LIB names
 * another file).  The reference is single-threaded C on one core and has no counterpart; a caller of the reference would run
 * one rkFD per instance (reference example/chain/boxdrop_test.c:22-54).
 * ndev <= 0: all visible devices; devices: their HIP ordinals, or NULL for 0 .. ndev-1.  Calls return 0 / negative like the
 * rkfdBatch calls (rkfdHipLastError names the device); rkfdNodeStatus returns the largest rkfdBatchStatus of the devices. */
typedef struct rkfdNode rkfdNode;
rkfdNode *rkfdNodeCreate(const rkfdModel *m, int total, int max_rigid, int ndev, const int *devices);
void rkfdNodeDestroy(rkfdNode *n);
int rkfdNodeDevices(const rkfdNode *n);
int rkfdNodeSize(const rkfdNode *n);
/* shard k: its HIP device and its block of instances [lo, hi); the rkfdBatch behind it (for the per-batch accessors) */
int rkfdNodeShard(const rkfdNode *n, int k, int *device, int *lo, int *hi);
rkfdBatch *rkfdNodeBatch(rkfdNode *n, int k);
/* host arrays over ALL instances, [total][ndof] / [total][nlink]: scattered to / collected from the devices' shards */
int rkfdNodeSetState(rkfdNode *n, const double *dis, const double *vel);
int rkfdNodeSetMotorInput(rkfdNode *n, const double *input);
int rkfdNodeGetState(rkfdNode *n, double *dis, double *vel, double *acc);      /* plain copies, no collective; any may be NULL */
/* rkfdBatchSpecialize / SetSplit / UpdateInit / Update / Snapshot / Restore / Status on every device, each issued by the
 * device's own host thread on the device's own stream; the call returns when every thread has issued its launches (the GPUs run
 * on; Status and Gather wait for them) */
int rkfdNodeSpecialize(rkfdNode *n);
int rkfdNodeSetSplit(rkfdNode *n, int nsplit);
int rkfdNodeSetStepsPerLaunch(rkfdNode *n, int steps);        /* rkfdBatchSetStepsPerLaunch on every device */
int rkfdNodeTuneInstancesPerWave(rkfdNode *n, int nsteps);    /* rkfdBatchTuneInstancesPerWave on every device (each keeps what is faster there) */
int rkfdNodeUpdateInit(rkfdNode *n);
int rkfdNodeUpdate(rkfdNode *n, int nsteps);
int rkfdNodeSnapshot(rkfdNode *n);
int rkfdNodeRestore(rkfdNode *n);
int rkfdNodeStatus(rkfdNode *n);
/* one ncclAllGather of the final {dis, vel}: afterwards every device holds all `total` final states (rkfdNodeGatherDev: on
 * device k, [ndev][mx][2 ndof] doubles - block j = shard j's instances, dis | vel per instance, padded to the largest shard mx),
 * and dis / vel ([total][ndof], either may be NULL) receive them on the host in instance order */
int rkfdNodeGather(rkfdNode *n, double *dis, double *vel);
const double *rkfdNodeGatherDev(const rkfdNode *n, int k, int *mx);

#ifdef __cplusplus
}
#endif
#endif
