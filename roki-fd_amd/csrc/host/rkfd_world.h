/* rkfd_world.h - builds the flattened rkfdModel from registered chains.
 * Plays the role of the rkFD cell list + rkCD cell/pair registry of the
 * reference (reference src/rkfd_sim.c:188-209, rkCDChainReg / rkCDPairChainUnreg).
 */
#ifndef RKFD_WORLD_H
#define RKFD_WORLD_H

#include "rkfd_model.h"
#include "rkfd_ztk.h"

typedef struct rkfdWorld_ {
  int nchain;
  rkfdChainDesc **chain;
  unsigned char *noself;/* [nchain] 1 = the pairs between cells of this chain itself were unregistered (rkCDPairChainUnreg) */
  int nci;
  rkfdContactInfo *ci;
  rkfdContactInfo cidef;/* solver default contact info */
  /* built model (owned) */
  rkfdModel model;
  void *blob;           /* single allocation backing all model arrays */
  int built;
} rkfdWorld;

void rkfdWorldInit(rkfdWorld *w);
void rkfdWorldDestroy(rkfdWorld *w);
/* takes ownership of c; returns chain id or -1 */
int  rkfdWorldAddChain(rkfdWorld *w, rkfdChainDesc *c);
/* remove chain id (rkCDChainUnreg + cell removal, reference src/rkfd_sim.c:237-255); later chains move down by one */
void rkfdWorldRemoveChain(rkfdWorld *w, int chain);
/* rkCDPairChainUnreg [RoKi; UNVERIFIED-DEP, meaning taken from the reference's own usage]: drop the collision pairs whose
 * TWO cells both belong to chain id - its self-collision pairs, which registration forms by default between shapes on different
 * links of one chain (reference src/rkfd_sim.c:198; the "self collision" branch of src/rkfd_util.c:163-170 serves them).  Pairs
 * with other chains stay: reference example/chain/arm_box_test.c:39-49 registers arm, box and floor and only then calls this
 * for the arm, which must go on touching both. */
void rkfdWorldPairChainUnreg(rkfdWorld *w, int chain);
int  rkfdWorldSetContactInfo(rkfdWorld *w, const char *filename);
/* (re)build w->model; returns 0 on success */
int  rkfdWorldBuild(rkfdWorld *w);
/* offset of a chain's joints in the packed state and its dof count */
int  rkfdWorldChainDofOffset(const rkfdWorld *w, int chain);
int  rkfdWorldChainLinkOffset(const rkfdWorld *w, int chain);

#endif
