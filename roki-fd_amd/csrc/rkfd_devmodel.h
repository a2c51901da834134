/* rkfd_devmodel.h - device-side view of an rkfdModel plus the derived tables the
 * wave-per-instance kernel needs (levels, ancestor tables for pointer jumping,
 * child lists, per-candidate geometry).  Built on the host by rkfd_devmodel_build()
 * (rkfd_capi.hip) and passed to the kernel by value; every pointer is a device pointer
 * (or a host pointer under the lane emulator used by tests).
 */
#ifndef RKFD_DEVMODEL_H
#define RKFD_DEVMODEL_H

#define RKFD_WAVE        64
#define RKFD_MAX_LINK    64
#define RKFD_MAX_DOF     64
#define RKFD_MAX_CAND    64
#define RKFD_MAX_ROWS    64   /* 3 * (rigid contact vertices) handled by one wave */

typedef struct {
  int nlink, ndof, ncand;
  int nlevel;            /* number of tree levels (max depth + 1)                       */
  int nround;            /* pointer-jumping rounds = ceil(log2(nlevel))                 */
  int nci;
  int solver, max_iter;
  int maxrg;             /* capacity: rigid contact vertices solved per instance (3*maxrg <= 64) */
  double dt, fric_w;
  /* per link */
  const int *parent, *jtype, *dofoff, *mtype, *depth, *is_static;
  const double *org, *mass, *com, *inertia;
  const double *stiff, *visc, *coulomb, *sfric;
  const double *mot_k, *mot_admit, *mot_vmax, *mot_vmin, *mot_gear, *mot_inertia;
  const int *anc;        /* [nround][nlink]: ancestor 2^r levels up, -1 if none          */
  const int *level_off;  /* [nlevel+1]                                                   */
  const int *level_link; /* [nlink] links sorted by depth                                */
  const int *child_off;  /* [nlink+1]                                                    */
  const int *child_idx;  /* [nlink-#roots]                                               */
  const int *pathlink;   /* [nlink][nlevel]: ancestor of link at depth d (d<=depth)      */
  /* per candidate contact vertex */
  const int *cand_linkA, *cand_linkB, *cand_foff, *cand_nf, *cand_ci;
  const double *cand_vert; /* [ncand*3] vertex in link A's frame                         */
  const double *planes;    /* [nplane*4] in link B's frame                               */
  /* contact infos */
  const int *ci_type;
  const double *ci_sf, *ci_kf, *ci_k, *ci_l, *ci_e, *ci_v;
} rkfdDevModel;

/* per-batch state arrays, instance-major: x[b*stride + j] */
typedef struct {
  double *dis, *vel, *acc;       /* [B][ndof]   */
  double *motor_in;              /* [B][nlink]  */
  int    *piv_type;              /* [B][nlink]  */
  double *piv_prev;              /* [B][nlink]  */
  int    *cv_active, *cv_type;   /* [B][ncand]  */
  double *cv_ref, *cv_f;         /* [B][ncand*3]*/
  double *dbg;                   /* optional debug dump, may be NULL */
  int dbg_stride;
  int batch;
} rkfdDevState;

#endif
