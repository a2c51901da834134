/* rkfd_devmodel_host.h - host-side builder of the device model tables. */
#ifndef RKFD_DEVMODEL_HOST_H
#define RKFD_DEVMODEL_HOST_H

#include <stddef.h>
#include "rkfd_model.h"
#include "rkfd_devmodel.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  rkfdDevModel dm;   /* pointers into blob (host addresses) */
  void *blob;
  size_t bytes;
  size_t lds_bytes;  /* LDS one instance needs */
  /* stick anchors (rkCDVert _ref) live in the frame of the other cell's link.  On the device that is the
   * DEVICE link (rigidly attached model links are merged into it); at the boundary they are given in the
   * model link's frame, as in the reference.  ref_frame[12*j]: frame (R row-major, p) of candidate j's
   * other model link in its device link (inside blob). */
  const double *ref_frame;
  int ncand;
} rkfdDevModelHost;

/* anchors between the model link's frame (boundary) and the device link's frame (device state), in place:
 * n = batch * ncand anchors, candidate index = k % ncand */
void rkfd_ref_to_device(const rkfdDevModelHost *h, double *ref, size_t n);
void rkfd_ref_to_model(const rkfdDevModelHost *h, double *ref, size_t n);

/* max_rigid: capacity of rigid contact vertices solved per instance */
int  rkfd_devmodel_build(const rkfdModel *m, int max_rigid, rkfdDevModelHost *out, char *err, int errlen);
/* the same for a kernel with 8 / ngroup instances per wavefront: ngroup = 8 (one instance, the default) or 4 (two instances: the
 * sweep schedule takes four links of a level per iteration; the world must fit 32 lanes - links, joint coordinates, contact
 * slots - and use neither the Vert QP nor the Volume plugin, else the call fails with a message) */
int  rkfd_devmodel_build_w(const rkfdModel *m, int max_rigid, int ngroup, rkfdDevModelHost *out, char *err, int errlen);
void rkfd_devmodel_free(rkfdDevModelHost *h);
/* shift every pointer of dm from the blob at `from` to its copy at `to` */
void rkfd_devmodel_rebase(rkfdDevModel *dm, const void *from, const void *to);

#ifdef __cplusplus
}
#endif
#endif
