#include "mgo_env.h"
int mgo_mystery_create(mgo_env* e, int variant) { (void)e; (void)variant; return -1; }
