// fjgpu_tlas.h -- device-side build of the instance level of every group (fjgpu_tlas.hip)
#ifndef FJGPU_TLAS_H
#define FJGPU_TLAS_H

#include <string>
#include <vector>

#include "fjgpu_types.h"

// d_instances: the scene's instance table already on the device.  members: the instance indices of
// every group concatenated; group g owns member_count[g] of them from member_first[g].  On success
// *d_nodes (hipMalloc, owned by the caller) holds the threaded lists of all groups, group g's
// node_count[g] nodes from node_first[g] (skip links are indices into *d_nodes).  Returns 0, or -1
// with *err set and nothing left allocated.
int TlasBuildDevice(const DInstance *d_instances, const std::vector<int> &members, const std::vector<int> &member_first,
    const std::vector<int> &member_count, std::vector<int> *node_first, std::vector<int> *node_count, DTNode **d_nodes, std::string *err);

#endif
