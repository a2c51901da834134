/* roki_fd/roki_fd.h - so that a driver written for the reference (#include <roki_fd/roki_fd.h>, reference
 * include/roki_fd/roki_fd.h) compiles against this build with -I<repo>/include: everything is in roki_fd_amd.h. */
#ifndef ROKI_FD_ROKI_FD_H
#define ROKI_FD_ROKI_FD_H
#include <stdlib.h>
#include "../roki_fd_amd.h"
#endif
