#define BANET_BUILD_ID "c5a9043dde3d5b89"
