/* roki_fd_amd.h - host C API of the MI355X-native rkFDUpdate path.
 *
 * Mirrors the public interface of roki-fd for this path - same function names, argument
 * meaning and error behaviour as reference include/roki_fd/rkfd_sim.h:54-98,
 * rkfd_property.h:15-34 and rkfd_solver.h:23-60 - so that a driver written against the
 * reference (the drivers under reference example/chain/) reads the same here.  Everything below the API is
 * new: models are flattened into an rkfdModel and every evaluation runs on the GPU
 * (include/rkfd_hip.h); nothing is computed on the host.
 *
 * RoKi / ZM types the examples touch directly are provided in the minimal form the API
 * needs (zVec, rkChain, rkJoint); they are NOT RoKi and carry only what is listed here.
 */
#ifndef ROKI_FD_AMD_H
#define ROKI_FD_AMD_H

#include <stdio.h>
#include <stdbool.h>
#include "rkfd_model.h"
#include "rkfd_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- minimal ZM vector ------------------------------------------------------------ */
typedef struct { int size; double *buf; } zVecStruct;
typedef zVecStruct *zVec;
zVec zVecAlloc(int size);
void zVecFree(zVec v);
/* zVecFreeAtOnce( n, v1, ..., vn ), zRandInit() as used by reference example/chain/arm_box_test.c:85, boxdrop_hardsoft_test.c:18 */
void zVecFreeAtOnce(int n, ...);
void zRandInit(void);
double zRandF(double min, double max);
void zVecFPrint(FILE *fp, zVec v);
#define zVecSize(v)      ( (v)->size )
#define zVecSizeNC(v)    ( (v)->size )
#define zVecBuf(v)       ( (v)->buf )
#define zVecElemNC(v,i)  ( (v)->buf[i] )
#define zVecElem(v,i)    ( (v)->buf[i] )
#define zDeg2Rad(d)      ( (d) * 3.14159265358979323846 / 180.0 )
#define eprintf(...)     fprintf( stderr, __VA_ARGS__ )

/* ---- minimal RoKi chain / joint views ---------------------------------------------- */
struct _rkFD;
typedef struct _rkChain rkChain;
typedef struct { rkChain *chain; int link; } rkJoint;
struct _rkChain {
  struct _rkFD *fd;   /* owner */
  int id;             /* chain id in the world */
  int nlink, ndof;
  int link_off, dof_off;
  rkJoint *joint;     /* [nlink] */
};
int  rkChainJointSize(rkChain *c);
int  rkChainLinkNum(rkChain *c);
#define rkChainLinkJoint(c,i) ( &(c)->joint[i] )
/* joint displacement / velocity of every joint of the chain as of the last update */
void rkChainGetJointDisAll(rkChain *c, zVec dis);
void rkChainGetJointVelAll(rkChain *c, zVec vel);
/* 1-DoF accessors used by the example controllers (reference example/chain/arm_box_test.c:16-21) */
void rkJointGetDis(rkJoint *j, double *dis);
void rkJointGetVel(rkJoint *j, double *vel);
void rkJointMotorSetInput(rkJoint *j, double *input);

/* ---- rkFDPrp (reference include/roki_fd/rkfd_property.h:15-34) -------------------------- */
typedef struct {
  double dt;
  int pyramid;
  double friction_weight;
  int max_iter;
  double vel_eps;
} rkFDPrp;
#define rkFDPrpDT(p)             (p)->dt
#define rkFDPrpSetDT(f,t)             ( (f)->prp.dt = (t) )
#define rkFDPrpSetPyramid(f,n)        ( (f)->prp.pyramid = (n) )
#define rkFDPrpSetFrictionWeight(f,w) ( (f)->prp.friction_weight = (w) )
#define rkFDPrpSetMaxIter(f,i)        ( (f)->prp.max_iter = (i) )
#define rkFDPrpSetVelEps(f,e)         ( (f)->prp.vel_eps = (e) )

/* ---- contact solver plugin table (reference include/roki_fd/rkfd_solver.h:23-60) --------- */
typedef struct { int type; double sf, kf, k, l, e, v; } rkContactInfo;
typedef struct { void *world; } rkFDCD;
#define rkFDCDBase(c) (c)
struct _rkFDSolver;
typedef struct {
  void (*_defci)(struct _rkFDSolver*, rkContactInfo*);
  bool (*_init)(struct _rkFDSolver*);
  void (*_colchk)(struct _rkFDSolver*, bool);
  bool (*_update)(struct _rkFDSolver*, bool);
  void (*_update_ref)(struct _rkFDSolver*);
  void (*_destroy)(struct _rkFDSolver*);
} rkFDSolverCom;
typedef struct _rkFDSolver {
  void *prp;            /* malloc'ed by rkFDSolverCreate_<Type>, freed by rkFDSolverReset */
  rkFDSolverCom *com;   /* static lifetime */
  double t;
  rkFDPrp *fdprp;
  rkFDCD *cd;
  struct _rkFD *fd;     /* the chains of the reference's rkFDChainArray live in fd */
} rkFDSolver;
void rkFDSolverInit(rkFDSolver *solver);
void rkFDSolverReset(rkFDSolver *solver);
void rkFDSolverDestroy(rkFDSolver *solver);
#define rkFDSolverGetDefaultContactInfo(s,c) (s)->com->_defci(s,c)
#define rkFDSolverUpdateInit(s)              (s)->com->_init(s)
#define rkFDSolverColChk(s,b)                (s)->com->_colchk(s,b)
#define rkFDSolverUpdate(s,b)                (s)->com->_update(s,b)
#define rkFDSolverUpdatePrevDrivingTrq(s)    (s)->com->_update_ref(s)
#define rkFDSolverUpdateDestroy(s)           (s)->com->_destroy(s)
/* plugins with a device path.  Vert (the reference's default): penalty (ELASTIC) contacts + the rigid branch with
 * friction pyramids and the active-set QP (reference src/rkfd_vert.c:258-336); MLCP: penalty + rigid PGS.
 * Volume (what the reference's drivers select; reference src/rkfd_volume.c): penalty + rigid pairs of CONVEX shapes by their
 * intersection volumes - a 6-D wrench per pair from the active-set QP, centre-of-normal-force and friction fix-ups with
 * the simplex LPs (csrc/device/rkfd_dev_volume.h).  A world whose rigid pairs are not convex polyhedra with at most 64
 * faces together is refused by rkFDUpdateInit with a message - nothing is approximated silently.
 * The table is the reference's (same six entries, same call protocol); what differs: `fd` stands where the reference
 * has `rkFDChainArray chains`, _colchk / _update_ref do nothing (collision detection and the previous driving
 * torque are part of the device step) and rkFDUpdate launches the fused device step instead of walking the table per
 * evaluation - so a third-party plugin written against the reference's table cannot be plugged in here; the boundary
 * of this build is the rkFDUpdate level. */
rkFDSolver *rkFDSolverCreate_Vert(rkFDSolver *s);
rkFDSolver *rkFDSolverCreate_MLCP(rkFDSolver *s);
rkFDSolver *rkFDSolverCreate_Volume(rkFDSolver *s);

/* ---- rkFD (reference include/roki_fd/rkfd_sim.h:24-52) ----------------------------------- */
typedef struct _rkFDCell {
  rkChain chain;
  struct _rkFDCell *next;
  struct _zShape3D *shape;   /* [number of shapes of the chain] handles for the slide-mode calls */
} rkFDCell;
#define rkFDCellChain(c) ( &(c)->chain )

typedef struct _rkFD {
  double t;
  rkFDPrp prp;
  rkFDSolver solver;
  rkFDCell *list;        /* registration order */
  rkContactInfo cidef;
  rkFDCD cd;
  zVec dis, vel, acc;    /* total joint state (owned; re-allocated by (un)registration) */
  int size;
  void *impl;            /* world builder + device batch */
} rkFD;

#define rkFDTime(f)   (f)->t
#define rkFDDT(f)     (f)->prp.dt
#define rkFDGetPrp(f) ( &(f)->prp )

rkFD *rkFDCreate(rkFD *fd);
void rkFDDestroy(rkFD *fd);
/* reference include/roki_fd/rkfd_sim.h:61-63.  rkFDChainReg registers a CLONE of `chain`; chains exist
 * here only as the views of registered cells (rkFDCellChain), of this or another rkFD.  Packed-state
 * pointers are invalidated by (un)registration, as in the reference (src/rkfd_sim.c:72-110). */
rkFDCell *rkFDChainReg(rkFD *fd, rkChain *chain);
rkFDCell *rkFDChainRegFile(rkFD *fd, char filename[]);
bool rkFDChainUnreg(rkFD *fd, rkFDCell *cell);
/* slide mode of a collision cell (fake crawler; reference src/rkfd_sim.c:384-440, include/roki_fd/rkfd_sim.h:74-80).
 * zShape3D / rkCDCell are handles here: rkFDCellShape( cell, i ) names shape number i of the cell's chain (its order
 * in the ZTK file), the collision cell of a shape is the shape itself.  Axis in the link frame. */
typedef struct { double e[3]; } zVec3D;
typedef struct _zShape3D { struct _rkFDCell *cell; int index; } zShape3D;
typedef zShape3D rkCDCell;
zShape3D *rkFDCellShape(struct _rkFDCell *cell, int i);
int rkFDCellShapeNum(struct _rkFDCell *cell);
void rkFDCDCellSetSlideMode(rkCDCell *cell, bool mode);
void rkFDCDCellSetSlideVel(rkCDCell *cell, double vel);
void rkFDCDCellSetSlideAxis(rkCDCell *cell, zVec3D *axis);
rkCDCell *rkFDShape3DGetCDCell(struct _rkFD *fd, zShape3D *shape);
rkCDCell *rkFDShape3DSetSlideMode(struct _rkFD *fd, zShape3D *shape, bool mode);
rkCDCell *rkFDShape3DSetSlideVel(struct _rkFD *fd, zShape3D *shape, double vel);
rkCDCell *rkFDShape3DSetSlideAxis(struct _rkFD *fd, zShape3D *shape, zVec3D *axis);
/* all three at once, the shape named by its number */
bool rkFDCellSetSlide(struct _rkFDCell *cell, int shape, bool mode, double vel, const double axis[3]);
void rkFDChainSetDis(rkFDCell *lc, zVec dis);
void rkFDChainSetVel(rkFDCell *lc, zVec vel);
bool rkFDContactInfoScanFile(rkFD *fd, char filename[]);
/* rkCDPairChainUnreg( rkFDCDBase(&fd.cd), chain ) as in reference example/chain/boxdrop_test.c:37 */
void rkCDPairChainUnreg(rkFDCD *cd, rkChain *chain);

/* rkFDFK / rkFDUpdateRate / rkFDUpdateFKRate (reference src/rkfd_sim.c:344-384): in the reference they push a packed
 * state into the chains' link frames / rates on the host.  Link frames and rates live on the device here and are
 * recomputed from the packed state by every evaluation, so these calls set the packed state the next
 * rkFDUpdateInit / rkFDUpdate starts from (dis; vel and acc) - there is no host-side kinematics to refresh. */
void rkFDFK(rkFD *fd, zVec dis);
void rkFDUpdateRate(rkFD *fd, zVec vel, zVec acc);
void rkFDUpdateFKRate(rkFD *fd);
/* rkChainFPrintZTK of every registered chain at its current joint displacements (reference src/rkfd_sim.c:587-593) */
void rkFDPrint(rkFD *fd);
void rkFDFPrintZTK(FILE *fp, rkFD *fd);

/* The device step integrates with the reference's default scheme only (zODE2 "Regular" wrapper + Runge-Kutta-Gill,
 * reference src/rkfd_sim.c:46-47), which is what every driver of the reference assigns.  Asking for another one is
 * NOT ignored: a message goes to stderr, rkFDStatus reports -2 until a supported scheme is assigned, and
 * rkFDUpdateInit / rkFDUpdate refuse to run - a caller never gets RKG results under another integrator's name. */
void rkfd_ode2_assign(rkFD *fd, const char *what, const char *type);
#define rkFDODE2Assign(f,t)        rkfd_ode2_assign( f, "rkFDODE2Assign", #t )
#define rkFDODE2AssignRegular(f,t) rkfd_ode2_assign( f, "rkFDODE2AssignRegular", #t )

#define rkFDSetSolver(f,type) do{                                 \
    rkFDSolverReset( &(f)->solver );                              \
    rkFDSolverCreate_##type( &(f)->solver );                      \
    rkFDSolverGetDefaultContactInfo( &(f)->solver, &(f)->cidef ); \
  } while(0)

void rkFDUpdateInit(rkFD *fd);
rkFD *rkFDUpdate(rkFD *fd);
void rkFDUpdateDestroy(rkFD *fd);
rkFD *rkFDSolve(rkFD *fd);
/* 0 when the last update ran cleanly; see rkfdBatchStatus for the codes */
int rkFDStatus(rkFD *fd);

/* ---- batched extension ------------------------------------------------------------- */
/* the flattened world of fd (valid after rkFDUpdateInit or rkFDBuildModel) */
const rkfdModel *rkFDBuildModel(rkFD *fd);
/* B independent copies of fd's world on one GPU; the returned handle is driven with the
 * rkfdBatch* functions of rkfd_hip.h */
rkfdBatch *rkFDBatchCreate(rkFD *fd, int batch, int device, int max_rigid);

/* ---- flat loader entry points (for FFI users: ctypes / cgo / JNI) --------------------- */
/* the world builder behind rkFD: register ZTK chains, read the contact-info table, get the
 * flattened rkfdModel that rkfdBatchCreate consumes */
typedef struct rkfdWorld_ rkfdWorldHandle;
rkfdWorldHandle *rkfdWorldCreate(void);
void rkfdWorldFree(rkfdWorldHandle *w);
/* rkFDChainRegFile: returns the chain id or -1 */
int  rkfdWorldRegFile(rkfdWorldHandle *w, const char *filename);
/* rkFDContactInfoScanFile: 0 on success */
int  rkfdWorldSetContactInfo(rkfdWorldHandle *w, const char *filename);
/* rkCDPairChainUnreg */
void rkfdWorldPairChainUnreg(rkfdWorldHandle *w, int chain);
/* rkFDPrpSet* + rkFDSetSolver */
void rkfdWorldSetPrp(rkfdWorldHandle *w, double dt, double friction_weight, int max_iter, int solver);
void rkfdWorldSetPyramid(rkfdWorldHandle *w, int pyramid);
int  rkfdWorldSetSlide(rkfdWorldHandle *w, int chain, int shape, int mode, double vel, const double axis[3]);
const rkfdModel *rkfdWorldModel(rkfdWorldHandle *w);
int  rkfdWorldChainDofOffset(const rkfdWorldHandle *w, int chain);
int  rkfdWorldChainLinkOffset(const rkfdWorldHandle *w, int chain);
/* joint displacements of the chain's [roki::chain::init] section; returns the chain's dof */
int  rkfdWorldChainInitDis(const rkfdWorldHandle *w, int chain, double *dis);
/* the chain written back in ZTK format (rkChainFPrintZTK); dis may be NULL (the file's [roki::chain::init]); 0 / -1 */
int  rkfdWorldWriteZTK(const rkfdWorldHandle *w, int chain, const char *filename, const double *dis);

#ifdef __cplusplus
}
#endif
#endif
