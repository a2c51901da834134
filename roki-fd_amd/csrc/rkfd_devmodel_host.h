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
} rkfdDevModelHost;

/* max_rigid: capacity of rigid contact vertices solved per instance */
int  rkfd_devmodel_build(const rkfdModel *m, int max_rigid, rkfdDevModelHost *out, char *err, int errlen);
void rkfd_devmodel_free(rkfdDevModelHost *h);
/* shift every pointer of dm from the blob at `from` to its copy at `to` */
void rkfd_devmodel_rebase(rkfdDevModel *dm, const void *from, const void *to);

#ifdef __cplusplus
}
#endif
#endif
