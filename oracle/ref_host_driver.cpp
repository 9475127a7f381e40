// ref_host_driver.cpp -- headless driver of the REFERENCE's own host-side classes, compiled together with their sources where they
// lie under /root/reference (recipe: `make -C oracle ref_host`, output oracle/_ref/ref_host_dump; nothing of the reference is copied
// into this repository and the binary never leaves the build container).  Test infrastructure only: tests/golden/make_host_golden.py
// runs it and commits what it prints as tests/golden/host_ref.npz + host_ref.json, the vectors tests/test_host_reference.py holds
// octhost_* (csrc/host_runtime.cpp, host_recorder.cpp, host_settings.cpp) against.
//
// Reference classes driven (unchanged sources, this image's Qt 5.9.7 + moc / uic):
//   AcquisitionBuffer             octproz_devkit/src/acquisitionbuffer.cpp:33-92
//   VirtualOCTSystem              octproz_plugins/octproz-virtual-oct-system/src/virtualoctsystem.cpp:27-368 (three feeding modes)
//   Recorder                      octproz/src/recorder.cpp:31-152
//   OctAlgorithmParametersManager octproz/src/octalgorithmparametersmanager.cpp:12-100 (curve CSV load / save)
//   SettingsFileManager           octproz/src/settingsfilemanager.cpp:28-112 (QSettings INI groups)
// The consumer side of the ring is this file's own loop, written after Processing::slot_start (octproz/src/processing.cpp:176-218):
// processing.cpp itself needs OpenGL and the CUDA pipeline and is not built.
//
// usage: ref_host_dump <command> ...   (QT_QPA_PLATFORM=offscreen); every command prints one JSON object on stdout
#include <QApplication>
#include <QDir>
#include <QFile>
#include <QSettings>
#include <QStringList>
#include <QThread>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "octalgorithmparametersmanager.h"
#include "octproz_devkit.h"
#include "recorder.h"
#include "settingsfilemanager.h"
#include "virtualoctsystem.h"

namespace {

uint32_t crc32_of(const void* data, size_t n) {  // zlib's CRC-32 (the fixtures are compared with zlib.crc32)
	static uint32_t table[256];
	static bool init = false;
	if (!init) {
		for (uint32_t i = 0; i < 256; i++) {
			uint32_t c = i;
			for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
			table[i] = c;
		}
		init = true;
	}
	uint32_t c = 0xFFFFFFFFu;
	const unsigned char* p = static_cast<const unsigned char*>(data);
	for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
	return c ^ 0xFFFFFFFFu;
}

std::string jstr(const QString& s) {
	std::string out = "\"";
	const QByteArray u = s.toUtf8();
	for (char ch : u) {
		const unsigned char c = (unsigned char)ch;
		if (c == '"' || c == '\\') { out += '\\'; out += ch; }
		else if (c == '\n') out += "\\n";
		else if (c == '\r') out += "\\r";
		else if (c == '\t') out += "\\t";
		else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); out += b; }
		else out += ch;
	}
	return out + "\"";
}

std::string hexbytes(const QByteArray& b) {
	static const char* d = "0123456789abcdef";
	std::string s;
	for (char ch : b) { s += d[((unsigned char)ch) >> 4]; s += d[((unsigned char)ch) & 15]; }
	return s;
}

// ---------------------------------------------------------------- AcquisitionBuffer
int cmd_buffer(int argc, char** argv) {
	const unsigned cnt = argc > 2 ? (unsigned)atoi(argv[2]) : 2u;
	const size_t bytes = argc > 3 ? (size_t)atoll(argv[3]) : 4104u;
	AcquisitionBuffer b;
	printf("{\"initial\": {\"currIndex\": %d, \"bufferCnt\": %u, \"bytesPerBuffer\": %zu, \"slots\": %d}", b.currIndex, b.bufferCnt, b.bytesPerBuffer, b.bufferArray.size());
	const bool ok = b.allocateMemory(cnt, bytes);
	printf(", \"allocate\": {\"ok\": %s, \"bufferCnt\": %u, \"bytesPerBuffer\": %zu, \"currIndex\": %d, \"slots\": [", ok ? "true" : "false", b.bufferCnt, b.bytesPerBuffer, b.currIndex);
	for (int i = 0; i < b.bufferArray.size(); i++) {
		const unsigned char* p = static_cast<const unsigned char*>(b.bufferArray[i]);
		bool zero = true;
		for (size_t k = 0; k < bytes; k++) zero = zero && p[k] == 0;
		printf("%s{\"align128\": %d, \"zero\": %s, \"ready\": %s}", i ? ", " : "", (int)((uintptr_t)p % 128), zero ? "true" : "false", b.bufferReadyArray[i] ? "true" : "false");
	}
	printf("]}");
	b.bufferReadyArray[cnt - 1] = true;
	// a second allocation releases the first one and resets the flags (acquisitionbuffer.cpp:47, :74)
	const bool ok2 = b.allocateMemory(cnt + 1, bytes / 2);
	printf(", \"reallocate\": {\"ok\": %s, \"bufferCnt\": %u, \"bytesPerBuffer\": %zu, \"ready\": [", ok2 ? "true" : "false", b.bufferCnt, b.bytesPerBuffer);
	for (int i = 0; i < b.bufferReadyArray.size(); i++) printf("%s%s", i ? ", " : "", b.bufferReadyArray[i] ? "true" : "false");
	printf("]}");
	b.releaseMemory();
	printf(", \"release\": {\"slots\": %d, \"flags\": %d, \"bufferCnt\": %u, \"bytesPerBuffer\": %zu, \"currIndex\": %d}}\n", b.bufferArray.size(), b.bufferReadyArray.size(), b.bufferCnt,
	       b.bytesPerBuffer, b.currIndex);
	return 0;
}

// ---------------------------------------------------------------- VirtualOCTSystem + a consumer after Processing::slot_start
// vos <file> <bitDepth> <width> <height> <depth> <buffersPerVolume> <buffersFromFile> <bscanOffset> <copyFileToRam> <sync> <waitUs> <consume>
int cmd_vos(int argc, char** argv) {
	if (argc < 14) return 2;
	simulatorParams sp;
	sp.filePath = QString::fromLocal8Bit(argv[2]);
	sp.bitDepth = atoi(argv[3]); sp.width = atoi(argv[4]); sp.height = atoi(argv[5]); sp.depth = atoi(argv[6]);
	sp.buffersPerVolume = atoi(argv[7]); sp.buffersFromFile = atoi(argv[8]); sp.bscanOffset = atoi(argv[9]);
	sp.copyFileToRam = atoi(argv[10]) != 0; sp.syncWithProcessing = atoi(argv[11]) != 0; sp.waitTimeUs = atoi(argv[12]);
	const unsigned consume = (unsigned)atoi(argv[13]);

	VirtualOCTSystem sys;
	QStringList messages;
	QObject::connect(&sys, &Plugin::info, [&](QString s) { messages << ("info: " + s); });
	QObject::connect(&sys, &Plugin::error, [&](QString s) { messages << ("error: " + s); });
	std::atomic<int> started{0}, stopped{0};
	QObject::connect(&sys, &AcquisitionSystem::acquisitionStarted, &sys, [&](AcquisitionSystem*) { started++; }, Qt::DirectConnection);
	QObject::connect(&sys, &AcquisitionSystem::acquisitionStopped, &sys, [&]() { stopped++; }, Qt::DirectConnection);
	sys.slot_updateParams(sp);  // what the settings dialog's "apply" emits (virtualoctsystem.cpp:355-367)

	printf("{\"acquisitionParams\": {\"samplesPerLine\": %u, \"ascansPerBscan\": %u, \"bscansPerBuffer\": %u, \"buffersPerVolume\": %u, \"bitDepth\": %u}",
	       sys.params->params.samplesPerLine, sys.params->params.ascansPerBscan, sys.params->params.bscansPerBuffer, sys.params->params.buffersPerVolume, sys.params->params.bitDepth);

	sys.acqusitionRunning = false;
	std::thread producer([&]() { sys.startAcquisition(); });   // OCTproZ runs it on the acquisition QThread (octprozapp.cpp)
	// wait for acquisitionStarted (or a failed init: acquisitionStopped without a start)
	while (!started.load() && !stopped.load()) QThread::usleep(50);

	struct Rec { int slot; unsigned nr; uint32_t crc; size_t bytes; };
	std::vector<Rec> seen;
	size_t bytesPerBuffer = 0;
	int ringSlots = 0;
	if (started.load()) {
		AcquisitionBuffer* buffer = sys.buffer;
		bytesPerBuffer = buffer->bytesPerBuffer;
		ringSlots = buffer->bufferArray.size();
		const unsigned buffersPerVolume = (unsigned)sp.buffersPerVolume;
		unsigned currBufferNr = buffersPerVolume - 1;                       // processing.cpp:148
		while (sys.acqusitionRunning) {                                      // processing.cpp:176
			const int bufferPos = buffer->currIndex;
			if (bufferPos >= 0 && bufferPos < buffer->bufferReadyArray.size()) {
				if (buffer->bufferReadyArray[bufferPos]) {
					currBufferNr = (currBufferNr + 1) % buffersPerVolume;   // :181
					seen.push_back(Rec{bufferPos, currBufferNr, crc32_of(buffer->bufferArray[bufferPos], bytesPerBuffer), bytesPerBuffer});   // (octCudaPipeline here, :187)
					buffer->bufferReadyArray[bufferPos] = false;            // :191
					if (seen.size() >= consume) sys.stopAcquisition();
				}
			}
			QCoreApplication::processEvents();                              // :215
		}
	}
	producer.join();
	QCoreApplication::processEvents();
	printf(", \"started\": %s, \"bytesPerBuffer\": %zu, \"ringSlots\": %d, \"consumed\": [", started.load() ? "true" : "false", bytesPerBuffer, ringSlots);
	for (size_t i = 0; i < seen.size(); i++) printf("%s[%d, %u, %u]", i ? ", " : "", seen[i].slot, seen[i].nr, seen[i].crc);
	printf("], \"messages\": [");
	for (int i = 0; i < messages.size(); i++) printf("%s%s", i ? ", " : "", jstr(messages[i]).c_str());
	printf("], \"running_after_stop\": %s, \"slots_after_cleanup\": %d}\n", sys.acqusitionRunning ? "true" : "false", sys.buffer->bufferArray.size());
	return 0;
}

// ---------------------------------------------------------------- Recorder
// recorder <name> <savePath> <timestamp> <fileName|-> <bufferSizeInBytes> <buffersToRecord> <startWithFirstBuffer> <ops...>
//   ops: r<currentBufferNr> = slot_record of the next pattern buffer, a = slot_abortRecording, i = slot_init again
int cmd_recorder(int argc, char** argv) {
	if (argc < 9) return 2;
	Recorder rec(QString::fromLocal8Bit(argv[2]));
	OctAlgorithmParameters::RecordingParams rp;
	rp.savePath = QString::fromLocal8Bit(argv[3]);
	rp.timestamp = QString::fromLocal8Bit(argv[4]);
	rp.fileName = strcmp(argv[5], "-") ? QString::fromLocal8Bit(argv[5]) : QString("");
	rp.bufferSizeInBytes = (size_t)atoll(argv[6]);
	rp.buffersToRecord = (unsigned)atoi(argv[7]);
	rp.startWithFirstBuffer = atoi(argv[8]) != 0;
	rp.recordRaw = true; rp.recordProcessed = false; rp.recordScreenshot = false; rp.saveMetaData = false; rp.saveAs32bitFloat = false; rp.stopAfterRecord = false;
	rp.dataType = (OctAlgorithmParameters::DATA_TYPE)0;
	QStringList events;
	QObject::connect(&rec, &Recorder::info, [&](QString s) { events << ("info: " + s); });
	QObject::connect(&rec, &Recorder::error, [&](QString s) { events << ("error: " + s); });
	QObject::connect(&rec, &Recorder::recordingDone, [&]() { events << "recordingDone"; });
	QObject::connect(&rec, &Recorder::readyToRecord, [&](bool b) { events << (b ? "readyToRecord: true" : "readyToRecord: false"); });
	auto state = [&](const char* op) {
		printf("{\"op\": \"%s\", \"recordingEnabled\": %s, \"recordingFinished\": %s, \"isRecording\": %s, \"events\": [", op, rec.recordingEnabled ? "true" : "false",
		       rec.recordingFinished ? "true" : "false", rec.isRecording ? "true" : "false");
		for (int i = 0; i < events.size(); i++) printf("%s%s", i ? ", " : "", jstr(events[i]).c_str());
		printf("]}");
		events.clear();
	};
	printf("{\"steps\": [");
	state("new");
	rec.slot_init(rp);
	printf(", ");
	state("init");
	std::vector<unsigned char> buf(rp.bufferSizeInBytes ? rp.bufferSizeInBytes : 1);
	unsigned counter = 0;
	for (int k = 9; k < argc; k++) {
		printf(", ");
		if (argv[k][0] == 'r') {
			const unsigned nr = (unsigned)atoi(argv[k] + 1);
			for (size_t j = 0; j < buf.size(); j++) buf[j] = (unsigned char)((counter * 131u + j * 7u + (j >> 8)) & 0xFFu);  // pattern of make_host_golden.py
			counter++;
			rec.slot_record(buf.data(), 12, 0, 0, 0, 0, nr);
		} else if (argv[k][0] == 'a') {
			rec.slot_abortRecording();
		} else if (argv[k][0] == 'i') {
			rec.slot_init(rp);
		}
		state(argv[k]);
	}
	printf("], \"files\": [");
	QDir dir(rp.savePath);
	bool first = true;
	if (!rp.savePath.isEmpty() && dir.exists()) {
		for (const QString& name : dir.entryList(QDir::Files, QDir::Name)) {
			QFile f(dir.filePath(name));
			f.open(QIODevice::ReadOnly);
			const QByteArray all = f.readAll();
			printf("%s{\"name\": %s, \"bytes\": %d, \"crc32\": %u}", first ? "" : ", ", jstr(name).c_str(), all.size(), crc32_of(all.constData(), (size_t)all.size()));
			first = false;
		}
	}
	printf("]}\n");
	return 0;
}

// ---------------------------------------------------------------- curve CSV (OctAlgorithmParametersManager)
// csv_load <file> <resampling|background>        -> values the reference reads (float bits)
// csv_save <file> <resampling|background> <float bits ...>   -> the bytes the reference writes
int cmd_csv_load(int argc, char** argv) {
	if (argc < 4) return 2;
	OctAlgorithmParametersManager m;
	OctAlgorithmParameters* p = m.getParams();
	QStringList events;
	QObject::connect(&m, &OctAlgorithmParametersManager::info, [&](QString s) { events << ("info: " + s); });
	QObject::connect(&m, &OctAlgorithmParametersManager::error, [&](QString s) { events << ("error: " + s); });
	const bool res = !strcmp(argv[3], "resampling");
	const unsigned samplesBefore = p->samplesPerLine;
	if (res) m.loadCustomResamplingCurveFromFile(QString::fromLocal8Bit(argv[2]));
	else m.loadPostProcessBackgroundFromFile(QString::fromLocal8Bit(argv[2]));
	const float* c = res ? p->customResampleCurve : p->postProcessBackground;
	const int n = res ? p->customResampleCurveLength : (int)p->postProcessBackgroundLength;
	const bool loaded = !events.isEmpty() && events[0].startsWith("info");
	printf("{\"loaded\": %s, \"count\": %d, \"bits\": [", loaded ? "true" : "false", loaded ? n : 0);
	for (int i = 0; loaded && i < n; i++) { uint32_t u; memcpy(&u, &c[i], 4); printf("%s%u", i ? ", " : "", u); }
	printf("], \"samplesPerLine_before\": %u, \"samplesPerLine_after\": %u, \"events\": [", samplesBefore, p->samplesPerLine);
	for (int i = 0; i < events.size(); i++) printf("%s%s", i ? ", " : "", jstr(events[i].section(" File used", 0, 0)).c_str());
	printf("]}\n");
	return 0;
}
int cmd_csv_save(int argc, char** argv) {
	if (argc < 4) return 2;
	OctAlgorithmParametersManager m;
	OctAlgorithmParameters* p = m.getParams();
	std::vector<float> v;
	for (int k = 4; k < argc; k++) { const uint32_t u = (uint32_t)strtoul(argv[k], nullptr, 10); float f; memcpy(&f, &u, 4); v.push_back(f); }
	const bool res = !strcmp(argv[3], "resampling");
	if (res) { p->loadCustomResampleCurve(v.data(), (int)v.size()); m.saveCustomResamplingCurveToFile(QString::fromLocal8Bit(argv[2])); }
	else { p->loadPostProcessingBackground(v.data(), (int)v.size()); m.savePostProcessBackgroundToFile(QString::fromLocal8Bit(argv[2])); }
	QFile f(QString::fromLocal8Bit(argv[2]));
	f.open(QIODevice::ReadOnly);
	const QByteArray all = f.readAll();
	printf("{\"bytes\": %d, \"hex\": \"%s\"}\n", all.size(), hexbytes(all).c_str());
	return 0;
}

// ---------------------------------------------------------------- settings INI (SettingsFileManager over QSettings::IniFormat)
// ini_read <file> <group>                 -> {key: string value} as QSettings hands them to the application (QVariant::toString)
// ini_write <file> <timestamp|-> <group> <key=t:value ...>   t = b (bool) i (int) u (uint) d (double) f (float) s (string) -> the bytes written
int cmd_ini_read(int argc, char** argv) {
	if (argc < 4) return 2;
	SettingsFileManager m(QString::fromLocal8Bit(argv[2]));
	const QVariantMap map = m.getStoredSettings(QString::fromLocal8Bit(argv[3]));
	printf("{");
	bool first = true;
	for (auto it = map.constBegin(); it != map.constEnd(); ++it) {
		const QVariant& v = it.value();
		// a value with unescaped commas comes back as a QStringList (QSettings INI rule): joined the way QVariant::toString would not
		const QString s = v.type() == QVariant::StringList ? v.toStringList().join(",") : v.toString();
		printf("%s%s: {\"string\": %s, \"bool\": %s, \"int\": %d, \"uint\": %u, \"double\": %.17g, \"isList\": %s}", first ? "" : ", ", jstr(it.key()).c_str(), jstr(s).c_str(),
		       v.toBool() ? "true" : "false", v.toInt(), v.toUInt(), v.toDouble(), v.type() == QVariant::StringList ? "true" : "false");
		first = false;
	}
	printf("}\n");
	return 0;
}
int cmd_ini_write(int argc, char** argv) {
	if (argc < 5) return 2;
	const QString path = QString::fromLocal8Bit(argv[2]);
	QFile::remove(path);
	{
		SettingsFileManager m(path);
		if (strcmp(argv[3], "-")) m.setTimestamp(QString::fromLocal8Bit(argv[3]));
		QString group;
		QVariantMap map;
		auto flush = [&]() { if (!group.isEmpty()) m.storeSettings(group, map); map.clear(); };
		for (int k = 4; k < argc; k++) {
			const QString a = QString::fromLocal8Bit(argv[k]);
			if (a.startsWith("[")) { flush(); group = a.mid(1, a.size() - 2); continue; }
			const int eq = a.indexOf('=');
			const QString key = a.left(eq), val = a.mid(eq + 3);
			const QChar t = a[eq + 1];
			if (t == 'b') map.insert(key, val == "1");
			else if (t == 'i') map.insert(key, val.toInt());
			else if (t == 'u') map.insert(key, val.toUInt());
			else if (t == 'd') map.insert(key, val.toDouble());
			else if (t == 'f') map.insert(key, val.toFloat());
			else map.insert(key, val);
		}
		flush();
	}
	QFile f(path);
	f.open(QIODevice::ReadOnly);
	const QByteArray all = f.readAll();
	printf("{\"bytes\": %d, \"hex\": \"%s\"}\n", all.size(), hexbytes(all).c_str());
	return 0;
}

}  // namespace

int main(int argc, char** argv) {
	qputenv("QT_QPA_PLATFORM", "offscreen");
	qputenv("QT_LOGGING_RULES", "*.debug=false");
	QApplication app(argc, argv);   // the virtual OCT system owns a QDialog
	qRegisterMetaType<simulatorParams>("simulatorParams");
	if (argc < 2) return 2;
	const std::string c = argv[1];
	if (c == "buffer") return cmd_buffer(argc, argv);
	if (c == "vos") return cmd_vos(argc, argv);
	if (c == "recorder") return cmd_recorder(argc, argv);
	if (c == "csv_load") return cmd_csv_load(argc, argv);
	if (c == "csv_save") return cmd_csv_save(argc, argv);
	if (c == "ini_read") return cmd_ini_read(argc, argv);
	if (c == "ini_write") return cmd_ini_write(argc, argv);
	return 2;
}
