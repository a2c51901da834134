/* rkfd_ztk.h - minimal ZTK reader for the keys the hot path needs.
 *
 * The reference loads models with RoKi's rkChainReadZTK and
 * rkContactInfoArrayReadZTK (reference src/rkfd_sim.c:229,264); RoKi/Zeo/ZEDA are
 * not available here, so this is an independent reader of the same text
 * format, restricted to the tags/keys used by the models the path is run on:
 *   [roki::chain] name
 *   [zeo::shape]  name type(box|polyhedron|sphere|cylinder|cone) center depth width height radius div vert face
 *   [roki::motor] name type(dc|trq) motorconstant admittance maxvoltage minvoltage
 *                 gearratio rotorinertia gearinertia max min
 *   [roki::link]  name jointtype(fixed|revolute|prismatic|float|spherical|breakablefloat) mass stuff COM inertia frame
 *                 pos att parent shape motor stiffness viscosity coulomb staticfriction forcethreshold torquethreshold
 *   [roki::chain::init] joint
 *   [roki::contact] bind staticfriction kineticfriction compensation relaxation
 *                 elasticity viscosity
 * Other tags ([zeo::optic] ...) and keys are skipped.  The curved primitives (sphere, cylinder, cone) become the convex
 * polyhedra Zeo's zShape3DToPH would make of them with `div` divisions (default 32) [UNVERIFIED-DEP: the exact vertex
 * placement of Zeo's tessellation]; `COM: auto` / `inertia: auto` are computed from the link's shapes (uniform density).
 * rkfdChainWriteZTK writes a chain back in the same format (the role of rkChainFPrintZTK, reference src/rkfd_sim.c:587-593).
 */
#ifndef RKFD_ZTK_H
#define RKFD_ZTK_H

#include <stdio.h>

#define RKFD_NAME_MAX 64

enum { RKFD_SHAPE_NONE = 0, RKFD_SHAPE_BOX = 1, RKFD_SHAPE_PH = 2, RKFD_SHAPE_SPHERE = 3, RKFD_SHAPE_CYLINDER = 4, RKFD_SHAPE_CONE = 5 };
typedef struct {
  char name[RKFD_NAME_MAX];
  int nvert;
  double *vert;      /* [nvert*3] link frame */
  int nplane;
  double *plane;     /* [nplane*4] outward unit normal + offset */
  int nface;
  int *face;         /* [nface*3] triangles (vertex indices), kept for the writer and the auto mass properties */
  int convex;        /* 1: every vertex lies on the inner side of every face plane (what vertex collision AGAINST this shape assumes) */
  int ptype;         /* what the file said: RKFD_SHAPE_* */
  double prm[8];     /* box: center(3) depth width height; sphere: center(3) radius; cylinder / cone: two points (6) radius */
  int div;
  /* slide mode of the shape's collision cell (fake crawler; reference src/rkfd_sim.c:384-440): off by default */
  int slide_mode; double slide_vel, slide_axis[3];
} rkfdShape;

typedef struct {
  char name[RKFD_NAME_MAX];
  int type;          /* RKFD_MOTOR_* */
  double k, admit, vmax, vmin, gear, rotor_inertia, gear_inertia;
} rkfdMotor;

typedef struct {
  char name[RKFD_NAME_MAX];
  char stuff[RKFD_NAME_MAX];
  char parent_name[RKFD_NAME_MAX];
  char motor_name[RKFD_NAME_MAX];
  int parent;        /* index within the chain, -1 = root */
  int jtype;
  double mass, com[3], inertia[9];
  double org[12];    /* R(9 row-major) p(3) */
  double stiff, visc, coulomb, sfric;
  double ep_f, ep_t; /* breakable float joint: force / torque thresholds */
  int auto_com, auto_inertia;   /* "COM: auto" / "inertia: auto" were given (resolved when the chain is read) */
  int motor;         /* index into chain motors, -1 none */
  int nshape;
  int shape[8];      /* indices into chain shapes */
} rkfdLinkDesc;

typedef struct {
  char name[RKFD_NAME_MAX];
  int nlink;  rkfdLinkDesc *link;
  int nshape; rkfdShape *shape;
  int nmotor; rkfdMotor *motor;
  int ndof;
  double *init_dis;  /* [ndof] from [roki::chain::init] (zeros when absent) */
} rkfdChainDesc;

typedef struct {
  char stuff[2][RKFD_NAME_MAX];
  int type;          /* RKFD_CONTACT_* */
  double sf, kf, k, l, e, v;
} rkfdContactInfo;

/* returns NULL on failure (message on stderr), like rkChainReadZTK */
rkfdChainDesc *rkfdChainReadZTK(const char *filename);
void rkfdChainDescFree(rkfdChainDesc *c);
/* writes the chain in ZTK format; dis (may be NULL) goes into [roki::chain::init].  The role of rkChainFPrintZTK. */
void rkfdChainWriteZTK(FILE *fp, const rkfdChainDesc *c, const double *dis);
/* deep copy (the role of rkChainClone in reference src/rkfd_sim.c:217); NULL on allocation failure */
rkfdChainDesc *rkfdChainDescClone(const rkfdChainDesc *c);

/* returns number of entries read (>=0) or -1 on failure; *out is malloc'ed */
int rkfdContactInfoReadZTK(const char *filename, rkfdContactInfo **out);

#endif
