// io.cpp — the on-disk formats either side of the path (include/mulls_hip.h "mulls_io_*"):
// KITTI .bin scans (DataIo::read_bin_file, include/common/dataio.hpp:357-378), PCD v0.7 files of PointXYZINormal
// (read_pcd_file / write_pcd_file, :279-312, i.e. pcl::io::loadPCDFile / savePCDFileBinary / savePCDFile) and the
// odometry pose lines (write_lo_pose_overwrite / _append, :1896-1926).  Host code only: records are the 48-byte
// PointXYZINormal layout every other entry point takes.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/mulls_hip.h"

namespace
{
struct Rec // pcl::PointXYZINormal as the default constructor leaves it: data[3] = 1, everything else 0
{
	float x = 0, y = 0, z = 0, one = 1.0f;
	float nx = 0, ny = 0, nz = 0, pad1 = 0;
	float intensity = 0, curvature = 0, pad2 = 0, pad3 = 0;
};
static_assert(sizeof(Rec) == MULLS_POINT_BYTES, "48-byte point record");

// destination offset of a PCD field name inside Rec, or -1 (fields the point type does not have are skipped, like PCL)
int field_offset(const std::string &name)
{
	static const struct
	{
		const char *name;
		int off;
	} map[] = {{"x", 0}, {"y", 4}, {"z", 8}, {"normal_x", 16}, {"normal_y", 20}, {"normal_z", 24}, {"intensity", 32}, {"curvature", 36}};
	for (const auto &m : map)
		if (name == m.name)
			return m.off;
	return -1;
}

// handler of the entry points' function-try-blocks: nothing is thrown across the ABI
int io_caught() noexcept
{
	try
	{
		throw;
	}
	catch (const std::bad_alloc &)
	{
		return MULLS_E_NOMEM;
	}
	catch (...)
	{
		return MULLS_E_IO;
	}
}

int deliver(const std::vector<Rec> &v, void *pts, uint32_t cap, uint32_t *n)
{
	if (n)
		*n = (uint32_t)v.size();
	const size_t k = std::min<size_t>(v.size(), cap);
	if (k && !pts)
		return MULLS_E_INVALID;
	if (k)
		std::memcpy(pts, v.data(), k * sizeof(Rec));
	return MULLS_OK;
}
} // namespace

extern "C"
{
	int mulls_io_read_kitti_bin(const char *path, void *pts, uint32_t cap, uint32_t *n)
	try
	{
		if (!path)
			return MULLS_E_INVALID;
		std::ifstream in(path, std::ios::in | std::ios::binary);
		if (!in.good())
			return MULLS_E_IO;
		std::vector<Rec> v;
		// the reference's loop pushes a point per iteration and tests the stream only afterwards: the failed read at the end
		// of the file appends one default-constructed point (all zero) — kept, callers index the cloud by position
		while (in.good() && !in.eof())
		{
			Rec p;
			in.read(reinterpret_cast<char *>(&p.x), 3 * sizeof(float));
			in.read(reinterpret_cast<char *>(&p.intensity), sizeof(float));
			p.intensity *= 255;
			v.push_back(p);
		}
		return deliver(v, pts, cap, n);
	}
	catch (...)
	{
		return io_caught(); // nothing is thrown across the ABI
	}

	int mulls_io_read_pcd(const char *path, void *pts, uint32_t cap, uint32_t *n)
	try
	{
		if (!path)
			return MULLS_E_INVALID;
		std::ifstream in(path, std::ios::in | std::ios::binary);
		if (!in.good())
			return MULLS_E_IO;
		std::vector<std::string> fields;
		std::vector<int> sizes, counts;
		std::vector<char> types;
		size_t points = 0, width = 0, height = 1;
		bool have_points = false;
		std::string data_kind, line;
		while (std::getline(in, line))
		{
			if (!line.empty() && line.back() == '\r')
				line.pop_back();
			if (line.empty() || line[0] == '#')
				continue;
			std::istringstream ss(line);
			std::string key;
			ss >> key;
			if (key == "FIELDS" || key == "COLUMNS")
				for (std::string f; ss >> f;)
					fields.push_back(f);
			else if (key == "SIZE")
				for (int s; ss >> s;)
					sizes.push_back(s);
			else if (key == "TYPE")
				for (char t; ss >> t;)
					types.push_back(t);
			else if (key == "COUNT")
				for (int c; ss >> c;)
					counts.push_back(c);
			else if (key == "WIDTH")
				ss >> width;
			else if (key == "HEIGHT")
				ss >> height;
			else if (key == "POINTS")
			{
				ss >> points;
				have_points = true;
			}
			else if (key == "DATA")
			{
				ss >> data_kind;
				break;
			}
		}
		if (fields.empty() || sizes.size() != fields.size() || types.size() != fields.size() || data_kind.empty())
			return MULLS_E_IO;
		if (counts.empty())
			counts.assign(fields.size(), 1);
		if (counts.size() != fields.size())
			return MULLS_E_IO;
		// The header is untrusted input: PCD v0.7 field sizes are 1, 2, 4 or 8 bytes and counts are positive; nothing is
		// allocated before the record size and the point count have been checked against what the file can hold.
		for (size_t f = 0; f < fields.size(); f++)
			if ((sizes[f] != 1 && sizes[f] != 2 && sizes[f] != 4 && sizes[f] != 8) || counts[f] < 1 || counts[f] > (1 << 16))
				return MULLS_E_IO;
		if (!have_points)
		{
			if (height != 0 && width > SIZE_MAX / height)
				return MULLS_E_IO;
			points = width * height;
		}
		size_t step = 0;
		std::vector<size_t> foff(fields.size());
		for (size_t f = 0; f < fields.size(); f++)
		{
			foff[f] = step;
			step += (size_t)sizes[f] * (size_t)counts[f];
		}
		const std::streamoff data_begin = in.tellg();
		in.seekg(0, std::ios::end);
		const std::streamoff file_end = in.tellg();
		in.seekg(data_begin);
		if (data_begin < 0 || file_end < data_begin || step == 0)
			return MULLS_E_IO;
		const size_t remaining = (size_t)(file_end - data_begin);
		// binary: points * step bytes follow; ascii: at least one character and a separator per value
		const size_t min_bytes_per_point = data_kind == "binary" ? step : 2 * fields.size() - 1;
		if (points > remaining / min_bytes_per_point || points > 0xffffffffu)
			return MULLS_E_IO;
		std::vector<Rec> v(points);
		if (data_kind == "binary")
		{
			std::vector<char> blob(points * step);
			in.read(blob.data(), (std::streamsize)blob.size());
			if ((size_t)in.gcount() != blob.size())
				return MULLS_E_IO;
			for (size_t f = 0; f < fields.size(); f++)
			{
				const int dst = field_offset(fields[f]);
				if (dst < 0 || types[f] != 'F' || sizes[f] != 4 || counts[f] != 1)
					continue; // PCL maps fields by name and requires the same datatype
				for (size_t i = 0; i < points; i++)
					std::memcpy(reinterpret_cast<char *>(&v[i]) + dst, blob.data() + i * step + foff[f], 4);
			}
		}
		else if (data_kind == "ascii")
		{
			for (size_t i = 0; i < points; i++)
			{
				if (!std::getline(in, line))
					return MULLS_E_IO;
				std::istringstream ss(line);
				for (size_t f = 0; f < fields.size(); f++)
					for (int c = 0; c < counts[f]; c++)
					{
						std::string tok;
						if (!(ss >> tok))
							return MULLS_E_IO;
						const int dst = field_offset(fields[f]);
						if (dst < 0 || types[f] != 'F' || sizes[f] != 4 || counts[f] != 1)
							continue;
						const float val = (tok == "nan" || tok == "-nan") ? std::nanf("") : std::strtof(tok.c_str(), nullptr);
						std::memcpy(reinterpret_cast<char *>(&v[i]) + dst, &val, 4);
					}
			}
		}
		else
			return MULLS_E_UNSUPPORTED; // binary_compressed (LZF): the reference never writes it
		return deliver(v, pts, cap, n);
	}
	catch (...)
	{
		return io_caught(); // nothing is thrown across the ABI
	}

	int mulls_io_write_pcd(const char *path, const void *pts, uint32_t n, uint32_t stride, int as_binary)
	try
	{
		if (!path || (n && !pts) || stride < MULLS_POINT_BYTES)
			return MULLS_E_INVALID;
		std::ofstream out(path, std::ios::out | std::ios::binary | std::ios::trunc);
		if (!out)
			return MULLS_E_IO;
		// write_pcd_file reshapes to width 1, height n before saving (dataio.hpp:290-292)
		out << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity normal_x normal_y normal_z curvature\n"
			<< "SIZE 4 4 4 4 4 4 4 4\nTYPE F F F F F F F F\nCOUNT 1 1 1 1 1 1 1 1\nWIDTH 1\nHEIGHT " << n << "\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n
			<< "\nDATA " << (as_binary ? "binary" : "ascii") << "\n";
		static const int order[8] = {0, 4, 8, 32, 16, 20, 24, 36};
		const unsigned char *p = static_cast<const unsigned char *>(pts);
		for (uint32_t i = 0; i < n; i++)
		{
			float f[8];
			for (int k = 0; k < 8; k++)
				std::memcpy(&f[k], p + (size_t)i * stride + order[k], 4);
			if (as_binary)
				out.write(reinterpret_cast<const char *>(f), sizeof(f));
			else
			{
				char buf[256];
				int len = 0;
				for (int k = 0; k < 8; k++)
					len += std::isnan(f[k]) ? std::snprintf(buf + len, sizeof(buf) - len, "%snan", k ? " " : "")
											: std::snprintf(buf + len, sizeof(buf) - len, "%s%.8g", k ? " " : "", (double)f[k]);
				out.write(buf, len);
				out.put('\n');
			}
		}
		out.close();
		return out ? MULLS_OK : MULLS_E_IO;
	}
	catch (...)
	{
		return io_caught(); // nothing is thrown across the ABI
	}

	int mulls_io_write_pose(const char *path, const double T[16], int append)
	try
	{
		if (!path || !T)
			return MULLS_E_INVALID;
		std::ofstream out(path, append ? std::ios::app : std::ios::out);
		if (!out)
			return MULLS_E_IO;
		out.precision(8); // setprecision(8), default float notation: %.8g
		for (int r = 0; r < 3; r++)
			for (int c = 0; c < 4; c++)
				out << T[r + 4 * c] << ((r == 2 && c == 3) ? "\n" : " ");
		out.close();
		return out ? MULLS_OK : MULLS_E_IO;
	}
	catch (...)
	{
		return io_caught(); // nothing is thrown across the ABI
	}
}
