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
  int *pair_off;        /* [nchain] chain's cells are not paired with chains whose bit is set ... see .c */
  unsigned char *nopair;/* [nchain*nchain] 1 = pairs between the two chains were unregistered */
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
/* drop every collision pair that involves chain id and a chain registered so far
 * (rkCDPairChainUnreg as used in reference example/chain/boxdrop_test.c:37) */
void rkfdWorldPairChainUnreg(rkfdWorld *w, int chain);
int  rkfdWorldSetContactInfo(rkfdWorld *w, const char *filename);
/* (re)build w->model; returns 0 on success */
int  rkfdWorldBuild(rkfdWorld *w);
/* offset of a chain's joints in the packed state and its dof count */
int  rkfdWorldChainDofOffset(const rkfdWorld *w, int chain);
int  rkfdWorldChainLinkOffset(const rkfdWorld *w, int chain);

#endif
