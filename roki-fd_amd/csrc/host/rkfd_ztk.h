/* rkfd_ztk.h - minimal ZTK reader for the keys the hot path needs.
 *
 * The reference loads models with RoKi's rkChainReadZTK and
 * rkContactInfoArrayReadZTK (reference src/rkfd_sim.c:229,264); RoKi/Zeo/ZEDA are
 * not available here, so this is an independent reader of the same text
 * format, restricted to the tags/keys used by the models the path is run on:
 *   [roki::chain] name
 *   [zeo::shape]  name type(box|polyhedron) center depth width height vert face
 *   [roki::motor] name type(dc|trq) motorconstant admittance maxvoltage minvoltage
 *                 gearratio rotorinertia gearinertia max min
 *   [roki::link]  name jointtype mass stuff COM inertia frame pos att parent shape motor
 *                 stiffness viscosity coulomb staticfriction
 *   [roki::chain::init] joint
 *   [roki::contact] bind staticfriction kineticfriction compensation relaxation
 *                 elasticity viscosity
 * Other tags ([zeo::optic] ...) and keys are skipped.  Non-convex or curved
 * shapes (cylinder, sphere, cone ...) are skipped with a note (out of scope).
 */
#ifndef RKFD_ZTK_H
#define RKFD_ZTK_H

#include <stdio.h>

#define RKFD_NAME_MAX 64

typedef struct {
  char name[RKFD_NAME_MAX];
  int nvert;
  double *vert;      /* [nvert*3] link frame */
  int nplane;
  double *plane;     /* [nplane*4] outward unit normal + offset */
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
/* deep copy (the role of rkChainClone in reference src/rkfd_sim.c:217); NULL on allocation failure */
rkfdChainDesc *rkfdChainDescClone(const rkfdChainDesc *c);

/* returns number of entries read (>=0) or -1 on failure; *out is malloc'ed */
int rkfdContactInfoReadZTK(const char *filename, rkfdContactInfo **out);

#endif
