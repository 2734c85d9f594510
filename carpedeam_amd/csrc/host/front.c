/* carpedeam (MI355X build), front end: the binary a deployment puts where the reference's `carpedeam` was.
 *
 *   carpedeam <module> <args>      modules of the hot path -> carpedeam_mi355x (the gfx950 host binary, host/main.cpp)
 *   carpedeam <anything else>      -> the reference's own binary ($CARPEDEAM_REF_BIN), started with argv[0] = THIS program
 *
 * Why argv[0]: the reference's main does setenv("MMSEQS", argv[0], true) (lib/mmseqs/src/commons/Application.cpp:198) and its
 * workflow scripts call every module as "$MMSEQS" <module> (data/nuclassemble.sh:105-136, data/guidedNuclAssemble.sh:35-201,
 * lib/mmseqs/data/workflow/linclust.sh:24-87).  With argv[0] pointing back here, `carpedeam ancient_assemble reads.fq out.fa tmp
 * --ancient-damage dhigh` runs the reference's own workflow drivers and scripts, and every kmermatcher / rescorediagonal /
 * ancient_correction / ancient_read_assemble / ancient_contig_merge / cyclecheck / createdb / createhdb / convert2fasta call of
 * those scripts comes back through this program and lands on the MI355X.  A module call whose flags the device path does not
 * implement (carpedeam_mi355x exits with status 77 before doing any work) is REFUSED: this program exits with EXIT_FAILURE and the
 * dispatch log gets a "refused <module>" line.  An owned module is never computed by the reference binary: this program has no code
 * path that starts the reference binary with an owned module (the opt-in hand-over of round 4, CARPEDEAM_ALLOW_REF_FALLBACK, is gone).
 *
 * This program never touches the GPU (it is plain C and links nothing of HIP), so it may exec; carpedeam_mi355x never execs.
 *
 *   CARPEDEAM_GPU_BIN        the device module binary   (default: carpedeam_mi355x next to this program)
 *   CARPEDEAM_REF_BIN        the reference binary       (default: none - unknown commands are "Invalid Command")
 *   CARPEDEAM_DISPATCH_LOG   append one line per call: "gpu|ref|refused <module>"
 */
#include <errno.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#define CDM_EXIT_UNSUPPORTED 77

static const char *const OWNED[] = {"kmermatcher", "rescorediagonal", "ancient_correction", "ancient_read_assemble", "ancient_contig_merge", "cyclecheck",
                                    "createdb", "createhdb", "convert2fasta", "ancient_reads_loop",
                                    /* the host-side modules of linclust's tail and the scripts' file modules (host/cluster.cpp) */
                                    "clust", "createsubdb", "filterdb", "mergeclusters", "result2repseq", "rmdb", "mvdb", "align", NULL};

static void logLine(const char *where, const char *module) {
    const char *p = getenv("CARPEDEAM_DISPATCH_LOG");
    if (!p || !*p) return;
    FILE *f = fopen(p, "a");
    if (!f) return;
    fprintf(f, "%s %s\n", where, module);
    fclose(f);
}

int main(int argc, char **argv) {
    char self[PATH_MAX], gpu[PATH_MAX + 32];
    ssize_t n = readlink("/proc/self/exe", self, sizeof(self) - 1);
    if (n <= 0) { fprintf(stderr, "carpedeam: can not resolve /proc/self/exe\n"); return EXIT_FAILURE; }
    self[n] = '\0';
    const char *g = getenv("CARPEDEAM_GPU_BIN");
    if (g && *g) snprintf(gpu, sizeof(gpu), "%s", g);
    else {
        snprintf(gpu, sizeof(gpu), "%s", self);
        char *slash = strrchr(gpu, '/');
        snprintf(slash ? slash + 1 : gpu, 32, "carpedeam_mi355x");
    }
    const char *ref = getenv("CARPEDEAM_REF_BIN");
    if (ref && !*ref) ref = NULL;
    if (argc < 2) {
        fprintf(stderr, "usage: carpedeam <command> [<args>]\n  on the MI355X:");
        for (int i = 0; OWNED[i]; i++) fprintf(stderr, " %s", OWNED[i]);
        fprintf(stderr, "\n  everything else (ancient_assemble, nuclassemble, linclust, ...): the reference binary named by CARPEDEAM_REF_BIN%s\n", ref ? "" : " (not set)");
        return EXIT_FAILURE;
    }
    int owned = 0;
    for (int i = 0; OWNED[i] && !owned; i++) owned = strcmp(argv[1], OWNED[i]) == 0;
    if (owned) {
        fflush(NULL);
        const pid_t pid = fork();
        if (pid < 0) { perror("carpedeam: fork"); return EXIT_FAILURE; }
        if (pid == 0) {
            argv[0] = gpu;
            execv(gpu, argv);
            fprintf(stderr, "carpedeam: can not start %s: %s\n", gpu, strerror(errno));
            _exit(127);
        }
        int st = 0;
        while (waitpid(pid, &st, 0) < 0) if (errno != EINTR) { perror("carpedeam: waitpid"); return EXIT_FAILURE; }
        if (WIFSIGNALED(st)) { logLine("gpu", argv[1]); return 128 + WTERMSIG(st); }
        if (WEXITSTATUS(st) != CDM_EXIT_UNSUPPORTED) { logLine("gpu", argv[1]); return WEXITSTATUS(st); }
        /* the refusal stands: an owned module is not computed by other code than the device path's */
        fprintf(stderr, "carpedeam: %s refused by the MI355X path (see the message above); not handed to the reference binary\n", argv[1]);
        logLine("refused", argv[1]);
        return EXIT_FAILURE;
    }
    {
        if (!ref) {
            fprintf(stderr, "Invalid Command: %s\n(the MI355X build implements the modules of the hot path; set CARPEDEAM_REF_BIN to the reference's binary and its workflows - "
                            "ancient_assemble, nuclassemble, linclust, ... - run with those modules on the device)\n", argv[1]);
            return EXIT_FAILURE;
        }
        logLine("ref", argv[1]);
    }
    argv[0] = self;            /* -> MMSEQS = this program (Application.cpp:198) */
    execv(ref, argv);
    fprintf(stderr, "carpedeam: can not start the reference binary %s: %s\n", ref, strerror(errno));
    return EXIT_FAILURE;
}
