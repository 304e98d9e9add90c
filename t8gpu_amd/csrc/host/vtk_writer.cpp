// vtk_writer.cpp -- SURVEY 8f-4: the on-disk side of save_variables_to_vtk / save_variable_to_vtk /
// save_mesh_to_vtk (t8gpu/mesh/mesh_manager.inl:588-623, subgrid_mesh_manager.inl:1051-1138,1185-1206).
// The reference hands its arrays to t8_forest_write_vtk_ext; t8code is not available here, so this writes
// the same thing directly: one VTK XML UnstructuredGrid piece per rank (.vtu: unshared corner points, one
// VTK_QUAD / VTK_HEXAHEDRON per leaf, cell fields treeid / mpirank / level / element_id + the user
// fields) and a .pvtu that ties the pieces together. A Subgrid mesh is written the way the reference
// does it: every block refined uniformly twice, its cells in z-order (data reordered on the device by
// t8gpu_hip_column_major_to_z_order).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "t8gpu_host.h"

namespace {

struct Appended {
  std::vector<char> bytes;
  template <class T>
  size_t add(const std::vector<T>& v) {
    const size_t   off = bytes.size();
    const uint64_t nb  = v.size() * sizeof(T);
    bytes.resize(off + 8 + nb);
    std::memcpy(bytes.data() + off, &nb, 8);
    if (nb) std::memcpy(bytes.data() + off + 8, v.data(), nb);
    return off;
  }
};

template <class T>
const char* vtk_type();
template <> const char* vtk_type<double>() { return "Float64"; }
template <> const char* vtk_type<int32_t>() { return "Int32"; }
template <> const char* vtk_type<int64_t>() { return "Int64"; }
template <> const char* vtk_type<uint8_t>() { return "UInt8"; }

template <class T>
void print_value(FILE* f, T v) { std::fprintf(f, "%lld ", static_cast<long long>(v)); }
template <>
void print_value<double>(FILE* f, double v) { std::fprintf(f, "%.17g ", v); }

template <class T>
void data_array(FILE* f, Appended& app, bool ascii, const char* name, int ncomp, const std::vector<T>& v) {
  std::fprintf(f, "        <DataArray type=\"%s\" Name=\"%s\" NumberOfComponents=\"%d\" ", vtk_type<T>(), name, ncomp);
  if (ascii) {
    std::fprintf(f, "format=\"ascii\">\n          ");
    for (size_t i = 0; i < v.size(); i++) {
      print_value<T>(f, v[i]);
      if (i % 12 == 11) std::fprintf(f, "\n          ");
    }
    std::fprintf(f, "\n        </DataArray>\n");
  } else {
    std::fprintf(f, "format=\"appended\" offset=\"%zu\"/>\n", app.add(v));
  }
}

}  // namespace

extern "C" int t8gpu_host_write_vtu(const char* path, int dim, int64_t num_elements, const double* centre, const int32_t* level,
                                    int cells_per_dim, int mpirank, int64_t first_element_id, int num_fields,
                                    const char* const* names, const int32_t* components, const double* const* data, int ascii) {
  if (!path || (dim != 2 && dim != 3) || num_elements < 0 || (cells_per_dim != 1 && cells_per_dim != 4) || num_fields < 0) return 1;
  if (num_elements > 0 && (!centre || !level)) return 1;
  for (int k = 0; k < num_fields; k++)
    if (!names || !names[k] || !components || (components[k] != 1 && components[k] != 3) || !data || !data[k]) return 1;
  const int     corners = 1 << dim;
  const int     sub     = cells_per_dim == 4 ? (dim == 3 ? 64 : 16) : 1;
  const int64_t ncell   = num_elements * sub;
  std::vector<double>  points(static_cast<size_t>(ncell) * corners * 3);
  std::vector<int64_t> conn(static_cast<size_t>(ncell) * corners), offs(ncell), ids(ncell);
  std::vector<uint8_t> types(ncell, dim == 3 ? 12 : 9);
  std::vector<int32_t> treeid(ncell, 0), ranks(ncell, mpirank), levels(ncell);
  // VTK_QUAD / VTK_HEXAHEDRON corner order (counter-clockwise bottom, then top)
  static const int cx[8] = {0, 1, 1, 0, 0, 1, 1, 0}, cy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, cz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
  for (int64_t e = 0; e < num_elements; e++) {
    const double h = 1.0 / static_cast<double>(1ll << level[e]);
    for (int m = 0; m < sub; m++) {
      // cell m of the block in the z-order of two uniform refinements -> (i, j, k)
      int ijk[3] = {0, 0, 0};
      if (sub > 1)
        for (int a = 0; a < dim; a++) ijk[a] = ((m >> a) & 1) | (((m >> (dim + a)) & 1) << 1);
      const double  hc = h / cells_per_dim;
      const int64_t c  = e * sub + m;
      double        lo[3];
      for (int a = 0; a < 3; a++) lo[a] = a < dim ? centre[3 * e + a] - 0.5 * h + ijk[a] * hc : 0.0;
      for (int v = 0; v < corners; v++) {
        double* p = &points[(static_cast<size_t>(c) * corners + v) * 3];
        p[0] = lo[0] + cx[v] * hc;
        p[1] = lo[1] + cy[v] * hc;
        p[2] = dim == 3 ? lo[2] + cz[v] * hc : 0.0;
        conn[static_cast<size_t>(c) * corners + v] = c * corners + v;
      }
      offs[c]   = (c + 1) * corners;
      ids[c]    = first_element_id * sub + c;
      levels[c] = level[e] + (sub > 1 ? 2 : 0);
    }
  }
  FILE* f = std::fopen(path, "wb");
  if (!f) return 2;
  Appended app;
  std::fprintf(f, "<?xml version=\"1.0\"?>\n<VTKFile type=\"UnstructuredGrid\" version=\"1.0\" byte_order=\"LittleEndian\" header_type=\"UInt64\">\n");
  std::fprintf(f, "  <UnstructuredGrid>\n    <Piece NumberOfPoints=\"%lld\" NumberOfCells=\"%lld\">\n", static_cast<long long>(ncell * corners),
               static_cast<long long>(ncell));
  std::fprintf(f, "      <Points>\n");
  data_array(f, app, ascii != 0, "Position", 3, points);
  std::fprintf(f, "      </Points>\n      <Cells>\n");
  data_array(f, app, ascii != 0, "connectivity", 1, conn);
  data_array(f, app, ascii != 0, "offsets", 1, offs);
  data_array(f, app, ascii != 0, "types", 1, types);
  std::fprintf(f, "      </Cells>\n      <CellData Scalars=\"treeid,mpirank,level,element_id\">\n");
  data_array(f, app, ascii != 0, "treeid", 1, treeid);
  data_array(f, app, ascii != 0, "mpirank", 1, ranks);
  data_array(f, app, ascii != 0, "level", 1, levels);
  data_array(f, app, ascii != 0, "element_id", 1, ids);
  for (int k = 0; k < num_fields; k++) {
    std::vector<double> v(data[k], data[k] + static_cast<size_t>(ncell) * components[k]);
    data_array(f, app, ascii != 0, names[k], components[k], v);
  }
  std::fprintf(f, "      </CellData>\n    </Piece>\n  </UnstructuredGrid>\n");
  if (!ascii) {
    std::fprintf(f, "  <AppendedData encoding=\"raw\">\n_");
    std::fwrite(app.bytes.data(), 1, app.bytes.size(), f);
    std::fprintf(f, "\n  </AppendedData>\n");
  }
  std::fprintf(f, "</VTKFile>\n");
  const bool ok = std::ferror(f) == 0;
  return (std::fclose(f) == 0 && ok) ? 0 : 3;
}

extern "C" int t8gpu_host_write_pvtu(const char* path, int num_pieces, const char* const* piece_files, int num_fields,
                                     const char* const* names, const int32_t* components) {
  if (!path || num_pieces < 1 || !piece_files || num_fields < 0) return 1;
  FILE* f = std::fopen(path, "wb");
  if (!f) return 2;
  std::fprintf(f, "<?xml version=\"1.0\"?>\n<VTKFile type=\"PUnstructuredGrid\" version=\"1.0\" byte_order=\"LittleEndian\" header_type=\"UInt64\">\n");
  std::fprintf(f, "  <PUnstructuredGrid GhostLevel=\"0\">\n    <PPoints>\n      <PDataArray type=\"Float64\" Name=\"Position\" NumberOfComponents=\"3\"/>\n    </PPoints>\n");
  std::fprintf(f, "    <PCellData Scalars=\"treeid,mpirank,level,element_id\">\n");
  std::fprintf(f, "      <PDataArray type=\"Int32\" Name=\"treeid\"/>\n      <PDataArray type=\"Int32\" Name=\"mpirank\"/>\n");
  std::fprintf(f, "      <PDataArray type=\"Int32\" Name=\"level\"/>\n      <PDataArray type=\"Int64\" Name=\"element_id\"/>\n");
  for (int k = 0; k < num_fields; k++)
    std::fprintf(f, "      <PDataArray type=\"Float64\" Name=\"%s\" NumberOfComponents=\"%d\"/>\n", names[k], components[k]);
  std::fprintf(f, "    </PCellData>\n");
  for (int p = 0; p < num_pieces; p++) std::fprintf(f, "    <Piece Source=\"%s\"/>\n", piece_files[p]);
  std::fprintf(f, "  </PUnstructuredGrid>\n</VTKFile>\n");
  return std::fclose(f) == 0 ? 0 : 3;
}
