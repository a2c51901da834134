/* rkfd_model.h - flattened, immutable description of one simulated "world"
 * (every chain registered in one rkFD), shared by the host C API, the HIP
 * device path and the CPU oracle.
 *
 * The reference keeps this information in pointer-linked RoKi objects
 * (rkChain / rkLink / rkJoint / rkMotor / rkCD cell+pair lists /
 * rkContactInfo array; see reference include/roki_fd/rkfd_sim.h:38-52 and
 * src/rkfd_sim.c:188-209).  Here it is a struct of plain arrays so that it
 * can be handed across a C ABI and copied to HBM once per batch.
 *
 * Conventions (documented in DESIGN.md, section "Model conventions"):
 *  - links are numbered so that parent[i] < i; parent[i] == -1 for a chain
 *    root, whose org frame is given in the world frame.
 *  - org[i] = (R row-major 9, p 3): frame of link i w.r.t. its parent at
 *    zero joint displacement.  Joint axis of revolute / prismatic = local z.
 *  - packed joint state (dis/vel/acc) concatenates links in index order,
 *    dofoff[i] .. dofoff[i]+jdof(i)-1; float joints carry
 *    (position 3, angle-axis 3) / (linear 3, angular 3) expressed in the org frame.
 *  - 6-D quantities are ordered (linear 3, angular 3) as in Zeo's zVec6D.
 */
#ifndef RKFD_MODEL_H
#define RKFD_MODEL_H

#ifdef __cplusplus
extern "C" {
#endif

/* SPHER: 3 DoF, displacement = angle-axis vector, rate = angular velocity, both in the joint-origin frame (the rotational half of
 * the float joint's convention).  BRFLOAT: RoKi's breakable float joint (reference example/model/wall.ztk:51-53, example/chain/arm_wall_test.c) - six
 * coordinates like a float joint; until it breaks the link is rigidly attached to its parent (coordinates and rates stay as
 * set, accelerations zero); at every COMMITTING evaluation (the reference runs rkChainUpdateABIWrench / ...CachedABIWrench only
 * there, src/rkfd_sim.c:507-514) the wrench the joint transmits is compared with the thresholds brk_f / brk_t (norm of the
 * force, norm of the torque about the link origin); beyond either the joint is broken from the next evaluation on and is a
 * float joint for good.  [UNVERIFIED-DEP: RoKi's rk_joint_brfloat is not here; DEVIATIONS.md] */
enum { RKFD_JOINT_FIXED = 0, RKFD_JOINT_REVOL = 1, RKFD_JOINT_PRISM = 2, RKFD_JOINT_FLOAT = 3, RKFD_JOINT_SPHER = 4, RKFD_JOINT_BRFLOAT = 5 };
enum { RKFD_MOTOR_NONE = 0, RKFD_MOTOR_TRQ = 1, RKFD_MOTOR_DC = 2 };
/* contact-info type, cf. RK_CONTACT_RIGID / RK_CONTACT_ELASTIC (reference src/rkfd_cd.c:39-46) */
enum { RKFD_CONTACT_RIGID = 0, RKFD_CONTACT_ELASTIC = 1 };
/* stick / slip state of a contact vertex or joint friction pivot
 * (RK_CONTACT_SF / RK_CONTACT_KF, reference src/rkfd_util.c:170,256,262) */
enum { RKFD_SF = 0, RKFD_KF = 1 };
/* contact solver selected with rkFDSetSolver (reference include/roki_fd/rkfd_sim.h:89-93) */
enum { RKFD_SOLVER_VERT = 0, RKFD_SOLVER_MLCP = 1, RKFD_SOLVER_VOLUME = 2 };

#define RKFD_G     9.80665   /* gravity, RoKi RK_G            [UNVERIFIED-DEP] */
#define RKFD_TOL   1.0e-12   /* ZM zTOL, used by zIsTiny etc. [UNVERIFIED-DEP] */

static inline int rkfd_joint_dof(int jtype)
{
  return ( jtype == RKFD_JOINT_FLOAT || jtype == RKFD_JOINT_BRFLOAT ) ? 6 : ( jtype == RKFD_JOINT_SPHER ? 3 : ( jtype == RKFD_JOINT_FIXED ? 0 : 1 ) );
}

typedef struct {
  /* ---- kinematic tree ------------------------------------------------ */
  int nlink;          /* number of links of all chains                     */
  int ndof;           /* packed joint size (rkFD.size)                      */
  int nchain;
  const int *parent;  /* [nlink]                                            */
  const int *jtype;   /* [nlink] RKFD_JOINT_*                               */
  const int *dofoff;  /* [nlink] offset into packed state                   */
  const int *chain;   /* [nlink] chain id                                   */
  const double *org;  /* [nlink*12]                                         */
  const double *mass; /* [nlink]                                            */
  const double *com;  /* [nlink*3]  link frame                              */
  const double *inertia; /* [nlink*9] about COM, link frame                 */
  /* ---- joint friction / motor (1-DoF joints; zero elsewhere) ---------- */
  const double *stiff, *visc, *coulomb, *sfric;        /* [nlink]           */
  const int *mtype;                                     /* [nlink] RKFD_MOTOR_* */
  const double *mot_k, *mot_admit, *mot_vmax, *mot_vmin, *mot_gear, *mot_inertia; /* [nlink] */
  /* ---- collision shapes: convex polyhedra in link frame --------------- */
  int nshape;
  const int *shape_link;   /* [nshape]                                      */
  const int *shape_voff;   /* [nshape+1] prefix offsets into verts          */
  const int *shape_foff;   /* [nshape+1] prefix offsets into planes         */
  const double *verts;     /* [nvert*3]                                     */
  const double *planes;    /* [nplane*4] outward unit normal n and offset d: inside <=> n.x - d <= 0 */
  /* slide mode of a shape's collision cell (rkFDCDCellSetSlideMode / Vel / Axis, reference src/rkfd_sim.c:384-401):
   * the surface is taken to run along itself like a crawler belt */
  const int *shape_slide_mode;      /* [nshape] 0 / 1                         */
  const double *shape_slide_vel;    /* [nshape]                               */
  const double *shape_slide_axis;   /* [nshape*3] in the link frame           */
  /* ---- collision pairs (rkCD plist) and contact infos ------------------ */
  int npair;
  const int *pair_shape;   /* [npair*2]                                     */
  const int *pair_ci;      /* [npair] index into ci_* arrays                */
  int nci;
  const int *ci_type;      /* [nci] RKFD_CONTACT_*                          */
  const double *ci_sf, *ci_kf;          /* static / kinetic friction coeff. */
  const double *ci_k, *ci_l;            /* RIGID: compensation, relaxation  */
  const double *ci_e, *ci_v;            /* ELASTIC: elasticity, viscosity   */
  /* ---- candidate contact vertices (derived from pairs) ----------------- */
  int ncand;
  const int *cand_pair;    /* [ncand] pair index                            */
  const int *cand_side;    /* [ncand] 0/1: which shape of the pair owns the vertex */
  const int *cand_vert;    /* [ncand] global vertex index                   */
  /* ---- step properties (rkFDPrp, reference include/roki_fd/rkfd_property.h:15-22) */
  double dt;
  double friction_weight;
  int max_iter;
  int solver;              /* RKFD_SOLVER_*                                 */
  int pyramid;             /* faces of the friction pyramid of the Vert plugin (rkFDPrp pyramid, default 8) */
  /* ---- breakable float joints (zero elsewhere) -------------------------- */
  const double *brk_f, *brk_t;   /* [nlink] force / torque threshold (ZTK forcethreshold / torquethreshold) */
} rkfdModel;

#ifdef __cplusplus
}
#endif
#endif /* RKFD_MODEL_H */
