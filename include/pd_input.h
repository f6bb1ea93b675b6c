/* source-compatibility shim: the reference ships this header (include/pd_input.h); the definitions live in cassie_io_types.h */
#include "cassie_io_types.h"
