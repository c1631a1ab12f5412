/* fj_multi_thread.h -- MtGetThreadID for plugin sources (reference src/fj_multi_thread.h:36): see
 * fj_plugin_abi.h. */
#ifndef FJ_MULTI_THREAD_H
#define FJ_MULTI_THREAD_H
#include "fj_plugin_abi.h"
#endif
