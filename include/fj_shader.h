/* fj_shader.h -- what a shader plugin source includes (reference src/fj_shader.h): everything it is
 * compiled against lives in fj_plugin_abi.h. */
#ifndef FJ_SHADER_H
#define FJ_SHADER_H
#include "fj_plugin_abi.h"
#endif
