/* rkfd_devmodel.h - device-side view of an rkfdModel plus the derived tables the
 * wave-per-instance kernel needs (levels, ancestor tables for pointer jumping,
 * child lists, per-candidate geometry).  Built on the host by rkfd_devmodel_build()
 * (rkfd_capi.hip) and passed to the kernel by value; every pointer is a device pointer
 * (or a host pointer under the lane emulator used by tests).
 */
#ifndef RKFD_DEVMODEL_H
#define RKFD_DEVMODEL_H

#define RKFD_WAVE        64
/* INSTANCES PER WAVEFRONT (round 3).  RKFD_W = 1: one instance has the wavefront's 64 lanes (every kernel of the library).
 * RKFD_W = 2: two instances share a wavefront, 32 lanes each - a build of the SAME device code in which LANE() counts within the
 * instance's half, loops stride over RKFD_WL lanes, BALLOT / BCAST act within the half, every LDS pointer and the instance index
 * are per-lane values, and the sweeps take four links of a level per iteration (the device model is built with ngroup = 4).  It exists as a world-specific
 * kernel only (rkfdBatchSpecialize with rkfdBatchSetInstancesPerWave( b, 2 )), for worlds whose links, joint coordinates and
 * contact capacity fit 32 lanes and that do not use the Vert QP or the Volume plugin. */
#ifndef RKFD_W
#define RKFD_W 1
#endif
#define RKFD_WL ( RKFD_WAVE/RKFD_W )      /* lanes of one instance */
#define RKFD_MAX_LINK    64
#define RKFD_MAX_DOF     64
#define RKFD_QP_NQ_MAX   24   /* Vert QP: up to this many unknowns (8 contact vertices) the factor of Q is kept in registers (rkfd_dev_vertqp.h) */
#define RKFD_MAX_CAND    4096 /* candidate contact vertices per instance (swept 64 at a time; 9 bytes of LDS each) */
#define RKFD_MAX_ROWS    128  /* 3 * (rigid contact vertices): two MLCP rows per lane at most */

typedef struct {
  int nlink, ndof, ncand;   /* nlink: device links (rigidly attached links are merged into their parents) */
  int nlink_model;          /* links of the rkfdModel: stride of motor_in / piv_* state arrays        */
  int nlevel;            /* number of tree levels (max depth + 1)                       */
  int nround;            /* pointer-jumping rounds = ceil(log2(nlevel))                 */
  int nci;
  int solver, max_iter;
  int maxact;            /* capacity: active contact vertices (rigid + elastic) per instance        */
  int pu_alias;          /* 1: the probe scratch PU fits in (and aliases) the C|PA block                */
  int anchor;            /* link whose position is the origin of the spatial (Pluecker) coordinates in every evaluation:
                            the first link that can move (-1: none, the world origin is used)           */
  int vert_rigid;        /* > 0: Vert plugin and rigid pairs exist -> the QP path and its LDS are set up;
                            2: at most RKFD_QP_NQ_MAX unknowns - the QP keeps the factor of Q in registers and needs no W block */
  int qscr_alias;        /* (unused since round 3, kept for the layout of the structure) */
  int ma_packed;         /* 1: the contact matrix is kept as a packed lower triangle (PGS kernels only), chosen where it lets one more instance share a CU */
  int ma_size;           /* doubles the contact matrix may take: full rows with an odd stride for the Vert QP, else the packed lower triangle */
  int pyramid;           /* faces of the Vert plugin's friction pyramid                                  */
  int npurow;            /* rows of PU per side: nlevel - pu_d0 (+6 with a float joint)                 */
  int pu_d0;             /* first tree level that holds a 1-DoF joint: the probe scratch has no rows above it     */
  int nside;             /* 1 when every rigid-capable pair has a static cell (probe walks one-sided) else 2 */
  int maxrg;             /* capacity: rigid contact vertices solved per instance (3*maxrg <= 128) */
  int mlcp_mfma;         /* kernel variants (name kept from round 2): bit 2 (on) the Vert QP's Q = A'A with v_mfma_f64_16x16x4_f64 - the one
                            product where the matrix cores pay; bit 3 the grouped Gauss-Seidel of rkfd_dev_mlcp.h switched off, bit 5 its
                            sweep-order matrix storage switched off (packed triangle instead) - A/B switches of the bit-identity tests,
                            set through rkfdDebugVariants, never from the environment */
  int has_brf;           /* the world holds breakable float joints (rkfd_dev_brf.h): device links with brf[] != 0 are float joints in the
                            tables and take the part of a fixed joint in every evaluation in which they are not broken */
  const int *brf;        /* [nlink] 1: the link hangs on a breakable float joint */
  const double *brk_f, *brk_t;   /* [nlink] its force / torque thresholds */
  int lds_poison;        /* debugging switch RKFD_DEBUG_POISON_LDS=1: 4-byte words of LDS every instance fills with all ones (NaN / -1) before it
                            starts, so that a read of storage nobody wrote shows in the results whatever ran on the CU before; 0: off */
  /* Volume plugin (solver == RKFD_SOLVER_VOLUME and rigid pairs exist; device/rkfd_dev_volume.h) */
  int vol_npair;         /* rigid pairs of the model */
  int vol_np;            /* capacity: pairs in collision at once (6 vol_np <= 64 unknowns)                     */
  int vol_ncp;           /* capacity: contact-plane conditions per pair (vol_np ( 1 + vol_ncp ) <= 64 constraints) */
  int vol_pv;            /* capacity: vertices of a clipped face polygon                                        */
  int vol_nf;            /* most faces the two shapes of a rigid pair have together (one lane each, <= 64)      */
  const int *vol_pair;   /* [vol_npair*8] device link A, B, contact info, first face loop of A, loops of A, first of B, loops of B,
                            slide mode (bit 0: cell[0], bit 1: cell[1]) */
  const int *vol_loop;   /* [nloop*2] first vertex, vertices of a face loop (counter-clockwise seen from outside) */
  const double *vol_lplane; /* [nloop*4] the loop's plane, device link frame                                    */
  const double *vol_lvert;  /* [nlv*3] loop vertices, device link frame; the loops of one shape are contiguous  */
  const double *vol_slide;  /* [vol_npair*16] per side: slide speed, axis (3), origin of the shape's model link (3), 0 - device link frame */
  double dt, fric_w;
  /* per link */
  const int *parent, *jtype, *dofoff, *mtype, *depth, *is_static;
  const int *dofkind;    /* [ndof] 1: first angular coordinate of a float joint, 2: the other two, else 0 */
  const int *orig;       /* [nlink] model link of a device link                                       */
  const double *org, *mass, *com, *inertia;
  const double *stiff, *visc, *coulomb, *sfric;
  const double *mot_k, *mot_admit, *mot_vmax, *mot_vmin, *mot_gear, *mot_inertia;   /* mot_inertia: reflected through the gear, 0 without a DC motor */
  const int *anc;        /* [nround][nlink]: ancestor 2^r levels up, -1 if none          */
  const int *level_off;  /* [nlevel+1]                                                   */
  const int *level_link; /* [nlink] links sorted by depth                                */
  const int *child_off;  /* [nlink+1]                                                    */
  const int *child_idx;  /* [nlink-#roots]                                               */
  const int *pathlink;   /* [nlink][nlevel]: ancestor of link at depth d (d<=depth); then [nlink]: top link (see host) */
  /* packed per-link ints (copied to LDS): see RKFD_LI_* */
  const int *linfo;      /* [nlink]                                                      */
  /* sweep schedule: (nsched+4) iterations x 8 lane groups x 4 ints
   *   {link (-1 none), linfo[link], nchild | flags<<8 | (pool slot+1)<<16 | (float slot+1)<<24, child_off[link]} */
  int nsched;
  const int *sched;
  int ngroup;            /* lane groups (links) per sweep iteration the schedule was built for: 8, or 4 for two instances per wavefront */
  int lds_instance;      /* bytes of LDS one instance owns (the second instance of a wavefront starts that far in) */
  int lds_shared;        /* two instances per wavefront: bytes of the world's static tables (candidate info, link info, child / pool
                            table, face offsets, path table) kept ONCE per wavefront behind the two instances' blocks; 0: every
                            instance keeps its own (one instance per wavefront; worlds with breakable joints, whose link info and
                            path tops are per instance) */
  int npool;             /* links whose articulated inertia must be staged in LDS for a gathering parent */
  int nfloat;            /* float joints (each owns a 6x6 Cholesky slot and a saved frame)          */
  const int *pslot;      /* [nlink] pool slot of the link, -1 when its Ia is handed over in registers */
  /* per candidate contact vertex */
  const int *cand_linkA, *cand_linkB, *cand_foff, *cand_nf, *cand_ci;
  const double *cand_vert; /* [ncand*3] vertex in link A's frame                         */
  const double *cand_bs;   /* [ncand*4] bounding sphere of the other shape in link B's frame: centre, squared radius (with a margin) */
  /* slide mode (fake crawler), only when has_slide: per candidate the two cells, owner first:
   * cs_mode [ncand*2] 0/1 | 2 when the anchor drift is expressed in the OWNER link's frame (reference index quirk,
   * src/rkfd_util.c:232), cs_par [ncand*14]: slide_vel, axis(3), origin of the shape's model link (3), both in the
   * device link's frame, owner then other */
  int has_slide;
  const int *cs_mode;
  const double *cs_par;
  const int *cinfo;        /* [ncand] packed: RKFD_CI_PACK                                     */
  const double *planes;    /* [nplane*4] in link B's frame                               */
  /* contact infos */
  const int *ci_type;
  const double *ci_sf, *ci_kf, *ci_k, *ci_l, *ci_e, *ci_v;
} rkfdDevModel;

/* DEVICE joint kinds beyond the model's fixed / revolute / prismatic / float (the 3-bit jt field of linfo).  A spherical joint of the
 * model becomes three device links: two massless pseudo-links and the real one, each a revolute joint about one axis (x, y, z)
 * of the joint-origin frame through the joint centre - three rank-1 eliminations in the sweeps equal the rank-3 one (block
 * elimination), the probes and the PGS see ordinary 1-DoF joints.  The axes are fixed in the PARENT-side frame and the whole
 * velocity-product term sits on the real link (rkfd_dev_kinematics.h). */
#define RKFD_DJT_SPHX 4
#define RKFD_DJT_SPHY 5
#define RKFD_DJT_SPHZ 6
/* LDS ints of the grouped Gauss-Seidel's remembered layout (rkfdLds.GC), kept by worlds with more than 16 rigid contact vertices (M = 3 maxrg rows) */
#define RKFD_GC_INTS 50
#define RKFD_GC_NEEDED(M) ( (M) > 3*16 )
#define RKFD_JT_IS1(jt) ( (jt) == RKFD_JOINT_REVOL || (jt) == RKFD_JOINT_PRISM || (jt) >= RKFD_DJT_SPHX )
#define RKFD_LI_PACK(par,jt,depth,stat,mt,off) \
  ( ((par)+1) | ((jt)<<8) | ((depth)<<11) | ((stat)<<18) | ((mt)<<19) | ((off)<<21) )
#define RKFD_LI_PAR(x)    ( ( (x) & 0xFF ) - 1 )
#define RKFD_LI_JT(x)     ( ( (x) >> 8 ) & 7 )
#define RKFD_LI_DEPTH(x)  ( ( (x) >> 11 ) & 0x7F )
#define RKFD_LI_STATIC(x) ( ( (x) >> 18 ) & 1 )
#define RKFD_LI_MT(x)     ( ( (x) >> 19 ) & 3 )
#define RKFD_LI_OFF(x)    ( ( (x) >> 21 ) & 0xFF )
/* packed candidate info: link A (7 bits), link B (7), contact info (6), faces of the other shape (12) */
#define RKFD_CI_PACK(a,b,ci,nf) ( (a) | ( (b) << 7 ) | ( (ci) << 14 ) | ( (int)( (unsigned)(nf) << 20 ) ) )
#define RKFD_CI_A(x)      ( (x) & 0x7F )
#define RKFD_CI_B(x)      ( ( (x) >> 7 ) & 0x7F )
#define RKFD_CI_CI(x)     ( ( (x) >> 14 ) & 0x3F )
#define RKFD_CI_NF(x)     ( (int)( (unsigned)(x) >> 20 ) )
#define RKFD_MAX_ROUND 6

/* doubles of the Volume plugin's LDS block (rkfd_lds_carve and rkfd_devmodel.cpp) */
/* kept from the collision phase to the solve: per pair 48, per condition 8; then the larger of the collision phase's scratch
 * (face polygons nf x pv x 3, reduction 16 nf + 16) and the solve's (Q n(n+1)/2, W n mc, S and EV mc^2 each, vectors 5n + mc + 64,
 * simplex 8 pyr ncp + 6), n = 6 np, mc = np ( 1 + ncp ) */
#define RKFD_VOL_LDS_COL(nf, pv) ( (nf)*(pv)*3 + 16*(nf) + 16 )
/* (the QP's arrays - VQL, VQW, VS, VEV, VQV - and the friction simplex's workspace VLP overlay each other: the wrenches have left
 *  the QP for the pair records before the first LP is set up) */
#define RKFD_VOL_LDS_QP(np, ncp) \
  ( ( 6*(np) )*( 6*(np)+1 )/2 + 6*(np)*(np)*( 1+(ncp) ) + 2*(np)*( 1+(ncp) )*(np)*( 1+(ncp) ) + ( 30*(np) + (np)*( 1+(ncp) ) ) )
#define RKFD_VOL_LDS_LP(ncp, pyr) ( 8*(pyr)*(ncp) + 6 )
#define RKFD_VOL_LDS_SOL(np, ncp, pyr) \
  ( RKFD_VOL_LDS_QP( np, ncp ) > RKFD_VOL_LDS_LP( ncp, pyr ) ? RKFD_VOL_LDS_QP( np, ncp ) : RKFD_VOL_LDS_LP( ncp, pyr ) )
#define RKFD_VOL_LDS_DOUBLES(np, ncp, pv, nf, pyr) \
  ( (np)*48 + (np)*(ncp)*8 + ( RKFD_VOL_LDS_COL( nf, pv ) > RKFD_VOL_LDS_SOL( np, ncp, pyr ) ? RKFD_VOL_LDS_COL( nf, pv ) : RKFD_VOL_LDS_SOL( np, ncp, pyr ) ) )


/* per-batch state arrays, instance-major: x[b*stride + j] */
typedef struct {
  double *dis, *vel, *acc;       /* [B][ndof]   */
  double *motor_in;              /* [B][nlink]  */
  int    *piv_type;              /* [B][nlink]  */
  double *piv_prev;              /* [B][nlink]  */
  int    *cv_active, *cv_type;   /* [B][ncand]  */
  double *cv_ref, *cv_f;         /* [B][ncand*3]*/
  int    *brk;                   /* [B][nlink]  breakable float joints: 1 once broken (state) */
  unsigned int *stat;            /* optional [B][4] running sums over the committing evaluations of rkFDUpdate steps:
                                    rigid contact vertices, elastic contact vertices, steps, unused; may be NULL */
  unsigned long long *prof;      /* optional [B][8] phase cycle counters, may be NULL */
  double *dbg;                   /* optional debug dump, may be NULL */
  int dbg_stride;
  int batch;
} rkfdDevState;

#endif
