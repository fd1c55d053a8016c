#define BANET_BUILD_ID "be40c2ceddc8e865"
