/* registration API of the host mirror without a GPU: rkFDChainRegFile / rkFDChainReg (clone) / rkFDChainUnreg
 * (reference src/rkfd_sim.c:188-255): packed-state bookkeeping and the rebuilt flat model */
#include <stdio.h>
#include <string.h>
#include "roki_fd_amd.h"
int main(int argc, char **argv){
  rkFD fd; rkFDCell *a, *b, *c, *d;
  char f1[256], f2[256], f3[256];
  snprintf(f1,256,"%s/box.ztk",argv[1]); snprintf(f2,256,"%s/floor.ztk",argv[1]); snprintf(f3,256,"%s/chain30.ztk",argv[1]);
  rkFDCreate(&fd);
  a = rkFDChainRegFile(&fd,f1); b = rkFDChainRegFile(&fd,f3); c = rkFDChainRegFile(&fd,f2);
  printf("size %d\n", fd.size);
  zVecElemNC(fd.dis, 6+3) = 0.25;     /* a joint of the chain */
  d = rkFDChainReg(&fd, rkFDCellChain(a));   /* clone of the box */
  printf("after clone size %d chain ids %d %d %d %d\n", fd.size, rkFDCellChain(a)->id, rkFDCellChain(b)->id, rkFDCellChain(c)->id, rkFDCellChain(d)->id);
  if( !rkFDChainUnreg(&fd, a) ) return 1;
  printf("after unreg size %d ids %d %d %d dofoff %d %d %d  q[3]=%g\n", fd.size, rkFDCellChain(b)->id, rkFDCellChain(c)->id, rkFDCellChain(d)->id,
    rkFDCellChain(b)->dof_off, rkFDCellChain(c)->dof_off, rkFDCellChain(d)->dof_off, zVecElemNC(fd.dis,3));
  const rkfdModel *m = rkFDBuildModel(&fd);
  printf("model nlink %d ndof %d ncand %d\n", m->nlink, m->ndof, m->ncand);
  if( rkFDChainUnreg(&fd, a) ) return 2;   /* no longer registered */
  rkFDDestroy(&fd);
  return 0;
}
