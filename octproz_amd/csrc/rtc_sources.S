/* rtc_sources.S -- the kernel headers as text inside the library: the sources hiprtc compiles for one samplesPerLine at
 * octpipe run time (mixedn_rtc.hip).  Assembled from csrc/ (the .incbin paths are relative to it). */
	.section .rodata
#define OCT_RTC_TEXT(sym, file) \
	.global sym; \
	.type sym, @object; \
sym: \
	.incbin file; \
	.byte 0; \
	.size sym, . - sym
OCT_RTC_TEXT(oct_rtc_src_kernels_h, "kernels.h")
OCT_RTC_TEXT(oct_rtc_src_fft_regs_h, "fft_regs.h")
OCT_RTC_TEXT(oct_rtc_src_mixedn_kernel_h, "mixedn_kernel.h")
OCT_RTC_TEXT(oct_rtc_src_mixedn_static_h, "mixedn_static.h")
OCT_RTC_TEXT(oct_rtc_src_mixedn_static_plan_h, "mixedn_static_plan.h")
	.section .note.GNU-stack,"",@progbits
