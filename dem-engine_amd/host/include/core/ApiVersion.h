// core/ApiVersion.h -- version macros of the reference's generated header (src/core/ApiVersion.h.in)
#pragma once
#define DEME_VERSION_MAJOR 2
#define DEME_VERSION_MINOR 1
#define DEME_VERSION_PATCH 0
#define DEME_API_VERSION "2.1.0-mi355x"
