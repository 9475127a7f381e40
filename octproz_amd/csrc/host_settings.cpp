// host_settings.cpp -- readers for the files OCTproZ itself writes, so that a reference
// installation's settings and calibration curves drive this pipeline unchanged:
//   * settings.ini (QSettings INI): groups [processing], [streaming], [record] and the acquisition
//     plug-in's group ("Virtual%20OCT%20System"); key names = the PROC_* / STREAM_* / REC_* macros of
//     octproz_project/octproz/src/sidebar.h:47-94; GUI -> parameter mapping as Sidebar::update*Params
//     (src/sidebar.cpp:319-430): spin-box doubles are assigned to float fields.
//   * curve CSV ("resampling.csv", "background.csv"): one header line, then "index;value" per line,
//     value = field 1 of a ';' split parsed as float (src/octalgorithmparametersmanager.cpp:12-30,
//     writer :32-45).
// No Qt: a small INI parser that understands QSettings' %XX escapes in group names and
// true/false booleans.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>

#include "../../include/octhost.h"

namespace {

std::string trim(const std::string& s) {
	size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
	return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

std::string percentDecode(const std::string& s) {  // QSettings escapes ' ' as %20 in group and key names
	std::string out;
	for (size_t i = 0; i < s.size(); ++i) {
		if (s[i] == '%' && i + 2 < s.size() + 0 && isxdigit((unsigned char)s[i + 1]) && isxdigit((unsigned char)s[i + 2])) {
			out.push_back((char)strtol(s.substr(i + 1, 2).c_str(), nullptr, 16));
			i += 2;
		} else {
			out.push_back(s[i]);
		}
	}
	return out;
}

typedef std::map<std::string, std::map<std::string, std::string>> Ini;

bool parseIni(const char* path, Ini& ini) {
	std::ifstream f(path);
	if (!f) return false;
	std::string line, group = "General";
	while (std::getline(f, line)) {
		line = trim(line);
		if (line.empty() || line[0] == ';' || line[0] == '#') continue;
		if (line.front() == '[' && line.back() == ']') { group = percentDecode(line.substr(1, line.size() - 2)); continue; }
		size_t eq = line.find('=');
		if (eq == std::string::npos) continue;
		std::string val = trim(line.substr(eq + 1));
		if (val.size() >= 2 && val.front() == '"' && val.back() == '"') val = val.substr(1, val.size() - 2);
		ini[group][percentDecode(trim(line.substr(0, eq)))] = val;
	}
	return true;
}

struct Group {
	const std::map<std::string, std::string>* m;
	bool has(const char* k) const { return m && m->count(k); }
	bool b(const char* k, bool d) const { if (!has(k)) return d; const std::string& v = m->at(k); return v == "true" || v == "1"; }
	double num(const char* k, double d) const { return has(k) ? atof(m->at(k).c_str()) : d; }
	std::string str(const char* k) const { return has(k) ? m->at(k) : std::string(); }
};

Group group(const Ini& ini, const char* name) {
	auto it = ini.find(name);
	return Group{it == ini.end() ? nullptr : &it->second};
}

void copyPath(char* dst, size_t n, const std::string& s) {
	if (!dst || n == 0) return;
	std::snprintf(dst, n, "%s", s.c_str());
}

}  // namespace

extern "C" {

int octhost_load_settings_ini(const char* path, OctPipeParams* params, OctHostCurveSettings* curves,
                              OctHostVirtualParams* vsys, char* vsysFilePath, size_t vsysFilePathSize) {
	if (!path || !params) return OCTPIPE_ERR_INVALID_ARGUMENT;
	Ini ini;
	if (!parseIni(path, ini)) return OCTPIPE_ERR_INVALID_ARGUMENT;
	const Group p = group(ini, "processing"), s = group(ini, "streaming"), r = group(ini, "record");
	// Sidebar::updateProcessingParams, sidebar.cpp:319-338
	params->bitshift = p.b("bitshift", params->bitshift);
	params->bscanFlip = p.b("flip_bscans", params->bscanFlip);
	params->signalLogScaling = p.b("log", params->signalLogScaling);
	params->signalGrayscaleMax = (float)p.num("max", params->signalGrayscaleMax);
	params->signalGrayscaleMin = (float)p.num("min", params->signalGrayscaleMin);
	params->signalMultiplicator = (float)p.num("coeff", params->signalMultiplicator);
	params->signalAddend = (float)p.num("addend", params->signalAddend);
	params->fixedPatternNoiseRemoval = p.b("fixed_pattern_removal", params->fixedPatternNoiseRemoval);
	params->continuousFixedPatternNoiseDetermination = p.b("fixed_pattern_removal_continuously", params->continuousFixedPatternNoiseDetermination);
	params->bscansForNoiseDetermination = (uint32_t)p.num("fixed_pattern_removal_bscans", params->bscansForNoiseDetermination);
	params->sinusoidalScanCorrection = p.b("sinusoidal_scan_correction", params->sinusoidalScanCorrection);
	params->rollingAverageWindowSize = (int32_t)p.num("background_removal_window_size", params->rollingAverageWindowSize);
	params->backgroundRemoval = p.b("background_removal", params->backgroundRemoval);
	params->postProcessBackgroundRemoval = p.b("post_processing_background_removal", params->postProcessBackgroundRemoval);
	params->postProcessBackgroundWeight = (float)p.num("post_processing_background_removal_weight", params->postProcessBackgroundWeight);
	params->postProcessBackgroundOffset = (float)p.num("post_processing_background_removal_offset", params->postProcessBackgroundOffset);
	// updateResamplingParams / updateDispersionParams / updateWindowingParams, sidebar.cpp:372-430
	params->resampling = p.b("resampling", params->resampling);
	params->resamplingInterpolation = (int32_t)p.num("resampling_interpolation", params->resamplingInterpolation);
	params->dispersionCompensation = p.b("dispersion_compensation", params->dispersionCompensation);
	params->windowing = p.b("windowing", params->windowing);
	// updateStreamingParams :340-345, updateRecordingParams :347-360
	params->streamToHost = s.b("streaming_enabled", params->streamToHost);
	params->streamingBuffersToSkip = (uint32_t)s.num("streaming_skip", params->streamingBuffersToSkip);
	params->streamFloatToHost = r.b("save_as_32_bit_float", params->streamFloatToHost);
	if (curves) {
		curves->c[0] = (float)p.num("resampling_c0", curves->c[0]); curves->c[1] = (float)p.num("resampling_c1", curves->c[1]);
		curves->c[2] = (float)p.num("resampling_c2", curves->c[2]); curves->c[3] = (float)p.num("resampling_c3", curves->c[3]);
		curves->d[0] = (float)p.num("dispersion_compensation_d0", curves->d[0]); curves->d[1] = (float)p.num("dispersion_compensation_d1", curves->d[1]);
		curves->d[2] = (float)p.num("dispersion_compensation_d2", curves->d[2]); curves->d[3] = (float)p.num("dispersion_compensation_d3", curves->d[3]);
		curves->windowType = (int32_t)p.num("window_type", curves->windowType);
		curves->windowCenter = (float)p.num("window_center_position", curves->windowCenter);
		curves->windowFillFactor = (float)p.num("window_fill_factor", curves->windowFillFactor);
		curves->customResampling = p.b("custom_resampling", curves->customResampling);
		copyPath(curves->customResamplingFilePath, sizeof(curves->customResamplingFilePath), p.str("custom_resampling_filepath"));
		copyPath(curves->postBackgroundFilePath, sizeof(curves->postBackgroundFilePath), p.str("post_processing_background_filepath"));
	}
	if (vsys) {  // VirtualOCTSystemSettingsDialog keys (virtualoctsystemsettingsdialog.cpp), group = plug-in name
		const Group v = group(ini, "Virtual OCT System");
		vsys->bitDepth = (unsigned)v.num("bit_depth", vsys->bitDepth);
		vsys->width = (unsigned)v.num("width", vsys->width);
		vsys->height = (unsigned)v.num("height", vsys->height);
		vsys->depth = (unsigned)v.num("depth", vsys->depth);
		vsys->buffersPerVolume = (unsigned)v.num("buffers_per_volume", vsys->buffersPerVolume);
		vsys->buffersFromFile = (unsigned)v.num("buffers_from_file", vsys->buffersFromFile);
		vsys->bscanOffset = (unsigned)v.num("bscan_offset", vsys->bscanOffset);
		vsys->waitTimeUs = (unsigned)v.num("wait_time", vsys->waitTimeUs);
		vsys->copyFileToRam = v.b("copy_file_to_ram", vsys->copyFileToRam != 0);
		vsys->syncWithProcessing = v.b("sync_with_processing", vsys->syncWithProcessing != 0);
		copyPath(vsysFilePath, vsysFilePathSize, v.str("file_path"));
	}
	return OCTPIPE_OK;
}

// Writer of the same file (the Recorder's "save meta info" leg stores the settings next to a recording, and OCTproZ reads the
// file back at start-up): the keys octhost_load_settings_ini understands, QSettings syntax (true / false, %20 in group names).
int octhost_save_settings_ini(const char* path, const OctPipeParams* params, const OctHostCurveSettings* curves,
                              const OctHostVirtualParams* vsys, const char* vsysFilePath, const char* timestamp) {
	if (!path || !params) return OCTPIPE_ERR_INVALID_ARGUMENT;
	FILE* f = fopen(path, "w");
	if (!f) return OCTPIPE_ERR_INVALID_ARGUMENT;
	auto B = [](int v) { return v ? "true" : "false"; };
	fprintf(f, "[General]\ntimestamp=%s\n\n", timestamp ? timestamp : "");
	fprintf(f, "[record]\nsave_as_32_bit_float=%s\n\n", B(params->streamFloatToHost));
	fprintf(f, "[processing]\n");
	fprintf(f, "addend=%.9g\nbitshift=%s\ncoeff=%.9g\n", params->signalAddend, B(params->bitshift), params->signalMultiplicator);
	fprintf(f, "dispersion_compensation=%s\n", B(params->dispersionCompensation));
	if (curves) for (int i = 0; i < 4; ++i) fprintf(f, "dispersion_compensation_d%d=%.9g\n", i, curves->d[i]);
	fprintf(f, "fixed_pattern_removal=%s\nfixed_pattern_removal_continuously=%s\nfixed_pattern_removal_bscans=%u\n", B(params->fixedPatternNoiseRemoval),
	        B(params->continuousFixedPatternNoiseDetermination), params->bscansForNoiseDetermination);
	fprintf(f, "flip_bscans=%s\nlog=%s\nmax=%.9g\nmin=%.9g\n", B(params->bscanFlip), B(params->signalLogScaling), params->signalGrayscaleMax, params->signalGrayscaleMin);
	fprintf(f, "resampling=%s\n", B(params->resampling));
	if (curves) for (int i = 0; i < 4; ++i) fprintf(f, "resampling_c%d=%.9g\n", i, curves->c[i]);
	fprintf(f, "resampling_interpolation=%d\nsinusoidal_scan_correction=%s\n", params->resamplingInterpolation, B(params->sinusoidalScanCorrection));
	if (curves) fprintf(f, "window_center_position=%.9g\nwindow_fill_factor=%.9g\nwindow_type=%d\n", curves->windowCenter, curves->windowFillFactor, curves->windowType);
	fprintf(f, "windowing=%s\nbackground_removal=%s\nbackground_removal_window_size=%d\n", B(params->windowing), B(params->backgroundRemoval), params->rollingAverageWindowSize);
	if (curves) fprintf(f, "custom_resampling=%s\ncustom_resampling_filepath=%s\npost_processing_background_filepath=%s\n", B(curves->customResampling),
	                    curves->customResamplingFilePath, curves->postBackgroundFilePath);
	fprintf(f, "post_processing_background_removal=%s\npost_processing_background_removal_offset=%.9g\npost_processing_background_removal_weight=%.9g\n\n",
	        B(params->postProcessBackgroundRemoval), params->postProcessBackgroundOffset, params->postProcessBackgroundWeight);
	fprintf(f, "[streaming]\nstreaming_enabled=%s\nstreaming_skip=%u\n\n", B(params->streamToHost), params->streamingBuffersToSkip);
	if (vsys) {
		fprintf(f, "[Virtual%%20OCT%%20System]\nbit_depth=%u\nbuffers_from_file=%u\nbuffers_per_volume=%u\ndepth=%u\nfile_path=%s\nheight=%u\nwait_time=%u\nwidth=%u\n",
		        vsys->bitDepth, vsys->buffersFromFile, vsys->buffersPerVolume, vsys->depth, vsysFilePath ? vsysFilePath : "", vsys->height, vsys->waitTimeUs, vsys->width);
		fprintf(f, "copy_file_to_ram=%s\nbscan_offset=%u\nsync_with_processing=%s\n", B(vsys->copyFileToRam), vsys->bscanOffset, B(vsys->syncWithProcessing));
	}
	fclose(f);
	return OCTPIPE_OK;
}

int octhost_load_curve_csv(const char* path, float* out, unsigned capacity, unsigned* count) {
	if (!path || !count) return OCTPIPE_ERR_INVALID_ARGUMENT;
	*count = 0;
	std::ifstream f(path);
	if (!f) return OCTPIPE_ERR_INVALID_ARGUMENT;  // the reference returns an empty curve (:17-20)
	std::string line;
	std::getline(f, line);  // header line is skipped unconditionally (:23)
	unsigned n = 0;
	while (std::getline(f, line)) {
		// QString::section(";", 1, 1): the text between the first and the second ';' (empty -> 0.0f)
		size_t a = line.find(';');
		std::string field;
		if (a != std::string::npos) {
			size_t b = line.find(';', a + 1);
			field = line.substr(a + 1, b == std::string::npos ? std::string::npos : b - a - 1);
		}
		if (out && n < capacity) out[n] = (float)atof(field.c_str());
		++n;
	}
	*count = n;
	return OCTPIPE_OK;
}

int octhost_save_curve_csv(const char* path, const float* curve, unsigned count) {
	if (!path || (!curve && count)) return OCTPIPE_ERR_INVALID_ARGUMENT;
	FILE* f = fopen(path, "w");
	if (!f) return OCTPIPE_ERR_INVALID_ARGUMENT;
	fprintf(f, "Sample Number;Sample Value\n");  // :35
	for (unsigned i = 0; i < count; ++i) fprintf(f, "%u;%.9g\n", i, curve[i]);
	fclose(f);
	return OCTPIPE_OK;
}

}  // extern "C"
