/* source-compatibility shim: the reference ships this header (include/cassie_out_t.h); the definitions live in cassie_io_types.h */
#include "cassie_io_types.h"
