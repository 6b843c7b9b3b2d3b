/* ABI version of libaqualora_hip.so: the one definition, shared by include/aqualora_hip.h (callers) and the library's own
 * aql_abi_version().  Bump when an EXISTING entry point changes its signature; new entry points do not need a bump. */
#ifndef AQUALORA_ABI_H
#define AQUALORA_ABI_H
#define AQL_ABI_VERSION 4
#endif
