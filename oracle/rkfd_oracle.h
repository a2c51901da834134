/* rkfd_oracle.h - CPU restatement (fp64, single thread) of the rkFDUpdate hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (roki-fd_amd/, include/)
 * may call or link this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / timed CPU stand-in.
 *
 * PARITY UNPINNED: the reference (mi-lib/roki-fd 1.7.9) ships no tests, golden
 * vectors or expected outputs (reference test/test.sh:5-13 globs *test.c, none exist),
 * and its arithmetic lives in un-vendored dependencies
 * (zeda=1.12.1; zm=1.14.5; zeo=1.20.8; roki=2.13.13 - reference libinfo:3) that are
 * absent here, so the reference cannot be built or run.  This restatement follows
 * the reference's own control flow (file:line cited per function in the .c) and the
 * published algorithms of the dependencies (Featherstone ABA, Runge-Kutta-Gill,
 * projected Gauss-Seidel); it is pinned only by the analytic known-answer tests in
 * tests/ and by self-generated golden vectors (labelled as such).
 */
#ifndef RKFD_ORACLE_H
#define RKFD_ORACLE_H

#include "rkfd_model.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rkfdOracle rkfdOracle;

rkfdOracle *rkfdOracleCreate(const rkfdModel *m);
void rkfdOracleDestroy(rkfdOracle *o);

void rkfdOracleSetState(rkfdOracle *o, const double *dis, const double *vel);
void rkfdOracleGetState(const rkfdOracle *o, double *dis, double *vel, double *acc);
void rkfdOracleSetMotorInput(rkfdOracle *o, const double *input /* [nlink] */);
double rkfdOracleTime(const rkfdOracle *o);

/* persistent contact-vertex state, one entry per model candidate vertex */
void rkfdOracleGetContact(const rkfdOracle *o, int *active, int *type, double *ref /*[ncand*3]*/, double *f /*[ncand*3]*/);
void rkfdOracleSetContact(rkfdOracle *o, const int *active, const int *type, const double *ref);
/* joint friction pivots, one entry per link (meaningful for 1-DoF joints) */
void rkfdOracleGetPivot(const rkfdOracle *o, int *type, double *prev_trq);
void rkfdOracleSetPivot(rkfdOracle *o, const int *type, const double *prev_trq);

/* breakable float joints: 1 per link whose joint has broken (state; 0 for every other link) */
void rkfdOracleGetBroken(const rkfdOracle *o, int *broken);
void rkfdOracleSetBroken(rkfdOracle *o, const int *broken);

/* rkFDUpdateInit (reference src/rkfd_sim.c:552-558): one committing evaluation at t */
void rkfdOracleUpdateInit(rkfdOracle *o);
/* rkFDUpdate (reference src/rkfd_sim.c:560-566): RKG stages + committing evaluation */
int  rkfdOracleUpdate(rkfdOracle *o);
/* test access to the building blocks of the Vert rigid branch: Moore-Penrose solve of a symmetric
 * system, and the active-set QP  min x'Qx/2 + c'x  s.t.  nf x >= d  started from unit normal forces
 * (constraint i acts on the three unknowns of contact i / P) */
void rkfdOraclePinvSolve(int n, const double *K, const double *rhs, double *x);
int  rkfdOracleQPASM(int n, int mc, int P, const double *q, const double *c, const double *nf, const double *d, double *ans, int *idx);
/* number of KKT solves of the last Vert QP (diagnostic) */
int  rkfdOracleLastQPIter(const rkfdOracle *o);
/* the last Vert QP (call with NULL data pointers for the sizes): min x'qx/2 + c'x s.t. nf x >= 0, result, active set */
void rkfdOracleGetLastQP(const rkfdOracle *o, int *n, int *mc, double *q, double *c, double *nf, double *ans, int *idx);
/* how many Vert QPs so far were ended by the circulation check instead of at the optimum (diagnostic) */
int  rkfdOracleQPCycleStops(const rkfdOracle *o);
/* Volume plugin: how often a GUARDED pair (a shape that is not convex: cannot be clipped) was found in collision so far */
int  rkfdOracleVolumeGuardHits(const rkfdOracle *o);
/* Volume plugin: number of colliding rigid pairs of the last evaluation; the data of one of them (see the .c); the
 * simplex LP min c'x s.t. Ax = b, x >= 0 (c NULL: feasibility only) */
int  rkfdOracleVolumePairs(const rkfdOracle *o);
int  rkfdOracleGetVolumePair(const rkfdOracle *o, int k, double *out, int cap);
int  rkfdOracleVolumeLP(int mr, int n, const double *A, const double *b, const double *c, double *x);
/* nsteps x rkFDUpdate */
int  rkfdOracleUpdateN(rkfdOracle *o, int nsteps);
/* bench.py's CPU baselines, threads and clock inside C (rkfd_oracle_mt.c): nthreads OS threads, each with its own oracle, doing
 * rollouts of `horizon` steps from the states dis / vel [ninst][ndof] (thread t takes instances t, t + nthreads, ...; horizon 0: one
 * trajectory of up to 1000 steps per instance) for `seconds`; returns the steps done, *elapsed = the longest thread's wall time */
long rkfdOracleRolloutsMT(const rkfdModel *m, int nthreads, int ninst, const double *dis, const double *vel, int horizon, double seconds, double *elapsed);
/* one dynamics evaluation at the current state: _rkFDUpdate / _rkFDUpdateRef
 * (reference src/rkfd_sim.c:533-549); result in acc */
int  rkfdOracleEval(rkfdOracle *o, int doUpRef);

/* introspection for the known-answer tests */
void rkfdOracleGetLinkFrames(const rkfdOracle *o, double *R /*[nlink*9]*/, double *p /*[nlink*3]*/);
void rkfdOracleGetLinkVelAcc(const rkfdOracle *o, double *vel /*[nlink*6]*/, double *acc /*[nlink*6]*/);
/* last MLCP system: returns number of rigid contact vertices nc; a is 3nc x 3nc row-major
 * (after relaxation was added), b after bias/compensation, f after division by dt */
int  rkfdOracleGetMLCP(const rkfdOracle *o, double *a, double *b, double *f, int cap);
/* count of floating-point operations is not instrumented; see DESIGN.md */

#ifdef __cplusplus
}
#endif
#endif
