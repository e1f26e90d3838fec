/* TEST INFRASTRUCTURE ONLY.  Instantiates the image decoder the reference vendors
 * (engine/stb_image.h, found through -I/root/reference/engine) so that the reference's
 * Texture::load_image / EnvironmentMap::load link inside oracle/_ref/ref_harness.
 * The reference's own engine/stb_impl.cpp also instantiates stb_image_write, whose
 * vendored copy needs the MSVC-only sprintf_s; the harness never writes images. */
#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"
