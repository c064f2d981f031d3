/* Forwarding stub: the reference splits its host API over AGAThA/src/host_batch.h and friends; here it is one header. */
#include "gasal_header.h"
