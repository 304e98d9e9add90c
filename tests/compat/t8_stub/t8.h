// TEST STUB -- not t8code, not product code. Opaque handles only, so that the reference's example translation
// units (which `#include <t8.h>`) can be COMPILED against include/t8gpu in an image without t8code
// (tests/test_reference_examples_compile.py: syntax / code generation only; nothing here can be linked or run).
#ifndef T8GPU_TEST_T8_STUB_H
#define T8GPU_TEST_T8_STUB_H
#include <cstdint>
typedef int32_t t8_locidx_t;
typedef struct t8_forest* t8_forest_t;
typedef struct t8_cmesh*  t8_cmesh_t;
struct t8_scheme_cxx;
typedef struct t8_scheme_cxx t8_scheme_cxx_t;
struct t8_element;
typedef struct t8_element t8_element_t;
#endif
