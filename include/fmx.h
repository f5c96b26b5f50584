/* placeholder — replaced below */
#ifndef FMX_H
#define FMX_H
#include <stdint.h>
#define FMX_OK 0
#define FMX_E_ARG (-1)
#endif
