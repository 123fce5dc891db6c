/*
 * opal_amd.h -- C ABI of the MI355X-native PIC step (libopal_amd.so).
 *
 * This is the drop-in boundary for opal's per-timestep loop: one flat function
 * per method of `trait Grid` (src/grid/mod.rs:47-120) and of `Population<T>`
 * (src/particle/mod.rs:147-376) that the time loop (src/main.rs:238-267) uses,
 * plus the host-side set-up the loop needs (GridDesign, the loader, the YAML
 * driver).  Opaque handles, plain pointers and sizes; no C++ or torch types.
 * INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add.
 *
 * Conventions
 *  - every function returning `int` returns an opal_status (0 = OK); on failure
 *    opal_last_error() holds a message for the calling thread.
 *  - the library owns all device memory.  Host arrays passed in/out are plain
 *    host pointers; nothing here takes or returns a device pointer except the
 *    exchange buffers of opal_link_t, which the transport owner asked for.
 *  - work is enqueued on the HIP stream given at creation (NULL = the default
 *    stream); calls that return values to the host synchronise that stream.
 *  - physics invariants that the reference enforces with `assert!` (particle
 *    moved >= 1 cell, electron.rs:313 / ion.rs:199; gather or deposit index out
 *    of the ghost range, yee.rs:501-502,578-579) are detected on the device and
 *    surface as OPAL_ERR_ASSERT from the next synchronising call.
 *  - calls on one handle are not re-entrant; `synchronize`, `advance` and the
 *    energy reductions are collective over subdomains exactly as in the
 *    reference ("must be called on all processes").
 */
#ifndef OPAL_AMD_H
#define OPAL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct opal_grid opal_grid_t; /* YeeGrid, src/grid/yee.rs:255-269 */
typedef struct opal_pop opal_pop_t;   /* Population<Electron|Ion>, src/particle/mod.rs:141-145 */
typedef struct opal_link opal_link_t; /* neighbour exchange; stands in for `impl Communicator` */
typedef struct opal_sim opal_sim_t;   /* main()'s state, src/main.rs:48-215 */
typedef struct opal_hist opal_hist_t; /* Histogram, src/particle/hgram.rs:71-84 */

/* enum Boundary, src/grid/mod.rs:22-28 */
enum opal_boundary { OPAL_BDY_INTERNAL = 0, OPAL_BDY_LASER = 1, OPAL_BDY_ABSORBING = 2, OPAL_BDY_CONDUCTING = 3 };
/* Electron = Vay push (electron.rs:268-330); Ion = Boris push with Z*, A (ion.rs:168-214) */
enum opal_species { OPAL_ELECTRON = 0, OPAL_ION = 1, OPAL_PHOTON = 2 };

enum opal_status {
    OPAL_OK = 0,
    OPAL_ERR_ARG = 1,      /* bad argument / handle */
    OPAL_ERR_HIP = 2,      /* HIP runtime error, or no GPU / extension unusable */
    OPAL_ERR_ASSERT = 3,   /* a reference `assert!` would have fired */
    OPAL_ERR_CAPACITY = 4, /* population or exchange buffer capacity exceeded */
    OPAL_ERR_CONFIG = 5,   /* ConfigError, src/setup.rs:33-61 */
    OPAL_ERR_IO = 6,
    OPAL_ERR_LINK = 7      /* exchange callback failed */
};

/* raw grid component ids for opal_grid_get_raw / set_raw (ghost-inclusive arrays) */
enum opal_comp {
    OPAL_X = 0, OPAL_RHO = 1, OPAL_JX = 2, OPAL_JY = 3, OPAL_JZ = 4,
    OPAL_EX = 5, OPAL_EY = 6, OPAL_EZ = 7, OPAL_BX = 8, OPAL_BY = 9, OPAL_BZ = 10
};

typedef double (*opal_fn1)(double x, void *user);
typedef double (*opal_fn2)(double t, double x, void *user);                    /* laser Ey/Ez, main.rs:89-90 */
typedef double (*opal_fn3)(double x, double urand, double nrand, void *user);  /* ux/uy/uz, main.rs:121-123 */

const char *opal_last_error(void);
const char *opal_version(void);
/* number of visible GPUs; OPAL_ERR_HIP when there is none */
int opal_device_count(int *count);
int opal_set_device(int device);
/* Which card a rank really runs on: the PCI bus id (hipDeviceGetPCIBusId, "0000:xx:00.0") and the device name of HIP
 * ordinal `device` (< 0: the current device).  No counterpart in the reference (MPI ranks are CPU processes); bench.py
 * gathers it over the ranks so that an N-GPU line shows N distinct cards whatever the transport. */
int opal_device_identity(int device, char *pci_bus_id, size_t len, char *name, size_t name_len);

/* ------------------------------------------------------------------ GridDesign
 * src/grid/mod.rs:126-151 (unbalanced) and :157-206 (balanced).  nx / offset
 * have `numtasks` entries.  balanced returns OPAL_ERR_CONFIG where the reference
 * panics on unwrap (:179). */
int opal_design_unbalanced(int numtasks, size_t size, double dx, size_t min_subsize, size_t *nx, double *offset);
int opal_design_balanced(int numtasks, size_t size, double xmin, double dx, size_t min_subsize,
                         opal_fn1 ne, void *user, size_t *nx, double *offset);

/* ------------------------------------------------------------------------ Grid */
/* Grid::build, grid/mod.rs:49, yee.rs:272-319 */
int opal_grid_build(opal_grid_t **out, int id, int numtasks, int left, int right, double xmin, double dx,
                    const size_t *nx, const double *offset, void *hip_stream);
int opal_grid_free(opal_grid_t *g);
int opal_grid_rank(const opal_grid_t *g);          /* yee.rs:321 */
int opal_grid_ngrids(const opal_grid_t *g);        /* yee.rs:325 */
/* A lone periodic subdomain is its own neighbour (yee.rs:365-370, mod.rs:313-326: self-copy).  By default
 * that is done in place; with on != 0 its halo and leavers go through the link instead (to itself), which
 * drives the whole multi-subdomain machinery -- packing, transport, unpacking -- on one GPU. */
int opal_grid_set_self_exchange(opal_grid_t *g, int on);
int opal_grid_to_left(const opal_grid_t *g);       /* yee.rs:329; -1 = None */
int opal_grid_to_right(const opal_grid_t *g);      /* yee.rs:337; -1 = None */
double opal_grid_xmin(const opal_grid_t *g);       /* yee.rs:531 */
double opal_grid_dx(const opal_grid_t *g);         /* yee.rs:535 */
size_t opal_grid_size(const opal_grid_t *g);       /* yee.rs:539: interior cells */
size_t opal_grid_len(const opal_grid_t *g);        /* interior + boundary cells */
size_t opal_grid_left_bdy_size(const opal_grid_t *g);
size_t opal_grid_right_bdy_size(const opal_grid_t *g);
size_t opal_grid_min_size(void);                   /* yee.rs:783 */
int opal_grid_clear(opal_grid_t *g);               /* yee.rs:543 */
/* yee.rs:551-641: adds the currents/charge of `pop` (needs prev_x, i.e. a prior opal_pop_advance) */
int opal_grid_deposit(opal_grid_t *g, const opal_pop_t *pop, double dt);
/* yee.rs:351-496: halo exchange through `link` (NULL = single subdomain), then
 * the laser / absorbing / conducting boundary conditions.  laser_y / laser_z are
 * evaluated on the host at exactly one (t, x), as the reference does (:458-461);
 * they may be NULL when the left boundary is not a laser. */
int opal_grid_synchronize(opal_grid_t *g, opal_link_t *link, opal_fn2 laser_y, opal_fn2 laser_z, void *user,
                          double t, double dt);
int opal_grid_advance(opal_grid_t *g, double dt);  /* yee.rs:345-349: B(dt/2), E(dt), B(dt/2) */
/* yee.rs:499-529 for n (cell, offset) pairs; E, B are [n][3] host arrays */
int opal_grid_fields_at(const opal_grid_t *g, size_t n, const int32_t *c, const double *xi, double *E, double *B);
/* yee.rs:787-809: this subdomain's part; the caller sums over subdomains (the MPI reduce) */
int opal_grid_em_field_energy(const opal_grid_t *g, double *energy);
/* yee.rs:815-835: `size` rows of 11 doubles (x rho jx jy jz Ex Ey Ez Bx By Bz), the payload of write_data */
int opal_grid_interpolate(const opal_grid_t *g, double *out);
/* yee.rs:749-781 for a single subdomain: writes "{dir}/{index}_grid.dat" */
int opal_grid_write_data(const opal_grid_t *g, const char *dir, size_t index);
/* the same from `n` gathered rows of opal_grid_interpolate (11 doubles each): what rank 0 writes, yee.rs:762-773 */
int opal_grid_write_rows(const double *rows, size_t n, const char *dir, size_t index);
/* ghost-inclusive component arrays (len doubles), for tests and for drivers that inject currents */
int opal_grid_get_raw(const opal_grid_t *g, int comp, double *out);
int opal_grid_set_raw(opal_grid_t *g, int comp, const double *in);
/* The paired fused push has an electrostatic form (Ey = Ez = B = 0 everywhere and uy = uz = 0 for every particle: it
 * neither streams uy, uz nor gathers more than Ex).  It is used while every upload was of that kind and until a push
 * meets a transverse field or momentum (each workgroup checks its tile and its population's flag; one that finds
 * something takes the general path, and the following launches are general).  This call withdraws what the grid has
 * heard so far: "the fields may be electrostatic again" -- results never depend on it, only the speed of the next
 * steps does (populations that have been through a general push stay general until their next upload). */
int opal_grid_assume_electrostatic(opal_grid_t *g);

/* ------------------------------------------------------------------ Population */
/* Population::new_empty (mod.rs:156); `capacity` slots are reserved on the device */
int opal_pop_create(opal_pop_t **out, int species, size_t capacity, void *hip_stream);
int opal_pop_free(opal_pop_t *p);
/* Population::new, mod.rs:172-203, with this library's documented RNG (xoshiro256**,
 * SplitMix64 seeding, seed = `seed`); draw order per particle as in the reference. */
int opal_pop_new(opal_pop_t **out, int species, size_t npc, opal_fn1 number_density, opal_fn3 ux, opal_fn3 uy,
                 opal_fn3 uz, void *user, const opal_grid_t *grid, uint64_t seed, double dt, void *hip_stream);
/* The same loader on the host only: fills caller arrays of `capacity` entries, *n_out = particle count */
int opal_loader_generate(int species, size_t npc, opal_fn1 number_density, opal_fn3 ux, opal_fn3 uy, opal_fn3 uz,
                         void *user, double grid_xmin, double dx, size_t ncells, uint64_t seed, size_t capacity,
                         size_t *n_out, int32_t *cell, double *x, double *pux, double *puy, double *puz, double *w);
/* Host threads the loaders use (Population::new is a serial loop in the reference, mod.rs:172-203; the particles are the
 * same bit for bit whatever the count: every particle takes a fixed number of draws from the one generator, so its state at
 * any cell is known).  0 (default): the cores there are, 16 at the most, for a Simulation's own expressions, and one for
 * opal_pop_new / opal_loader_generate, whose callbacks belong to the caller; n > 0: that many everywhere -- the callbacks
 * must then be callable from any thread. */
int opal_set_loader_threads(int n);
/* Ion::with_charge_state applied to every ion (main.rs:142-147, ion.rs:236-241) */
int opal_pop_set_charge_state(opal_pop_t *p, double z_star, double z, double a);
/* T::create for n particles (electron.rs:85-103, ion.rs:81-101): prev_x is derived from u, dx, dt */
int opal_pop_upload_soa(opal_pop_t *p, size_t n, const int32_t *cell, const double *x, const double *ux,
                        const double *uy, const double *uz, const double *w, double dx, double dt);
/* live particles only, in storage order; any pointer may be NULL.  prev_x is valid after advance. */
int opal_pop_download_soa(const opal_pop_t *p, int32_t *cell, double *x, double *prev_x, double *ux,
                          double *uy, double *uz, double *w);
int opal_pop_len(const opal_pop_t *p, size_t *n);  /* store.len() */
/* the same from the device's counters (slots in use minus slots vacated since the last re-binning): one small copy instead of
 * the whole cell array; exact between the library's own operators */
int opal_pop_live(opal_pop_t *p, size_t *n);
/* Population::advance, mod.rs:247-376: gather + push + migration to the neighbours
 * of `grid` through `link` (NULL = single subdomain: periodic self-wrap, or deletion
 * at open edges).  Stores prev_x for a later opal_grid_deposit. */
int opal_pop_advance(opal_pop_t *p, opal_link_t *link, const opal_grid_t *grid, double dt);
/* The fused hot path: advance + this population's deposit into `grid` in ONE pass
 * over the particles (same result as opal_pop_advance then opal_grid_deposit up to
 * floating-point summation order).  Particles deleted at an open edge deposit nothing,
 * as in the reference where deletion precedes deposition (main.rs:239-257). */
int opal_pop_advance_deposit(opal_pop_t *p, opal_link_t *link, opal_grid_t *grid, double dt);
/* mod.rs:227-240: this subdomain's part */
int opal_pop_total_kinetic_energy(const opal_pop_t *p, double *energy);
/* Re-bins the particles by cell (the reference sorts inside every advance, mod.rs:288).
 * Results never depend on it being called; throughput does. */
int opal_pop_sort(opal_pop_t *p, const opal_grid_t *grid);
/* 0 = never re-bin automatically; n = every n calls of advance; <0 = adaptive (default) */
int opal_pop_set_sort_interval(opal_pop_t *p, int interval);
/* diagnostics: out[0] slots in use, [1] particles deleted at open edges, [2] particle-steps that missed
 * their workgroup's LDS tile since the last re-binning, [3] steps since the last re-binning, [4] particles
 * per thread of the push kernels, [5] live particles and [6] non-empty cells at the last re-binning,
 * [7] host upper bound of the slots in use */
int opal_pop_counters(const opal_pop_t *p, uint32_t *out);
/* Measurement aid (default on).  With 0 the pushes stream every attribute of every particle whatever the data hold:
 * no electrostatic launches (opal_grid_assume_electrostatic) and no uniform weight taken from the launch arguments.
 * Results are the same either way; bench.py uses it to report the general kernel beside the default. */
int opal_set_elision(int on);
/* diagnostic: start != 0 begins recording the pushes' tile misses; start == 0 copies out[0] = misses seen and up to
 * 4096 x (workgroup, slot, cell, tile base cell) into out (1 + 4 * 4096 int32) and stops */
int opal_pop_debug_misses(opal_pop_t *p, int start, int32_t *out);
/* level 1: carry chi, work, y, z for electrons (electron.rs:18-25) / chi, y, z, parent id for photons;
 * level 2 (electrons): also the optical depth and id that photon emission needs; 0 = off (default) */
int opal_pop_track_extras(opal_pop_t *p, int on);
/* the same by field, so that a run carries only what its outputs read (each costs 8-16 B of HBM traffic per
 * particle-update): OPAL_TRACK_EMISSION = optical depth + id (+ chi), OPAL_TRACK_ID = a photon's parent id,
 * OPAL_TRACK_ABSORPTION (photons) = optical depths against absorption and stimulated emission + birth time */
enum { OPAL_TRACK_CHI = 1, OPAL_TRACK_WORK = 2, OPAL_TRACK_YZ = 4, OPAL_TRACK_EMISSION = 8, OPAL_TRACK_ID = 16,
       OPAL_TRACK_ABSORPTION = 32 };
int opal_pop_track_fields(opal_pop_t *p, unsigned fields);
/* opal_pop_advance stores prev_x for a later opal_grid_deposit (default); 0 = the caller never deposits this
 * population (photons; any species when current_deposition is off), so the 8 B per update are saved */
int opal_pop_keep_prev_x(opal_pop_t *p, int on);
int opal_pop_download_extras(const opal_pop_t *p, double *chi, double *work, double *y, double *z);
/* Optical depth against emission (electron.rs:24) and the id that keys the emission random numbers, for
 * electrons tracked at level 2; for photons (level >= 1) `id` is the parent electron's.  Slot order. */
int opal_pop_download_emission_state(const opal_pop_t *p, double *tau, uint64_t *id);
int opal_pop_upload_emission_state(opal_pop_t *p, size_t n, const double *tau, const uint64_t *id);

/* ------------------------------------------------------------------ Photon emission
 * emit_radiation (particle/interactions.rs:45-107): every electron whose optical depth has fallen below
 * zero emits a photon sampled from the quantum synchrotron spectrum (qed/photon_emission.rs:129-203),
 * recoils and draws a new optical depth (electron.rs:208-251); photons passing the filters are appended
 * to `photons`.  min_energy [MeV], max_angle [rad, to the negative x axis], max_formation_length [m]:
 * negative = no limit.  Random numbers: Philox4x32-10 keyed by (seed, electron id, step).
 * n_emitted (optional) receives the number of electrons that emitted. */
int opal_emit_radiation(opal_pop_t *electrons, opal_pop_t *photons, uint64_t seed, uint32_t step, double t,
                        double min_energy, double max_angle, double max_formation_length, size_t *n_emitted);
/* ------------------------------------------------------------------ Photon absorption
 * absorb (particle/interactions.rs:145-340): every photon meets the electrons of its own cell in turn
 * (Photon::interacts_with, photon.rs:312-362), running down its optical depths against one-photon absorption
 * (qed/photon_absorption.rs:16-34) and stimulated emission (qed/stimulated_emission.rs:16-37; Airy function
 * qed/special_functions/airy.rs:19-69) by w_e (c dt / dx) sigma per electron, and stops at the first that falls
 * below zero: an absorbed photon is removed and gives the electron its momentum (Electron::kick,
 * electron.rs:256-262); a stimulated emission appends a photon of the same momentum with the electron's weight
 * and takes that momentum from the electron.  max_displacement [m, transverse distance travelled since birth]
 * and stop_time [s since birth]: negative = no limit.  The electrons are re-binned by cell first (the
 * reference requires both populations sorted; the photons' order does not matter here).  Needs electrons that
 * track chi and ids (OPAL_TRACK_EMISSION or OPAL_TRACK_CHI | OPAL_TRACK_ID) and photons that track
 * OPAL_TRACK_CHI | OPAL_TRACK_YZ | OPAL_TRACK_ID | OPAL_TRACK_ABSORPTION.  Random numbers: Philox4x32-10 keyed by
 * (seed, photon's parent id and birth time, step).  n_absorbed / n_stimulated are optional. */
int opal_absorb(opal_pop_t *electrons, opal_pop_t *photons, const opal_grid_t *grid, uint64_t seed, uint32_t step,
                double t, double dt, double max_displacement, double stop_time, size_t *n_absorbed,
                size_t *n_stimulated);
/* A photon's optical depths against absorption / stimulated emission and its birth time, slot order (dead slots
 * skipped); any pointer may be null.  The upload overwrites the first n slots (call right after
 * opal_pop_upload_soa). */
int opal_pop_download_absorption_state(const opal_pop_t *p, double *tau_abs, double *tau_st, double *birth_time);
int opal_pop_upload_absorption_state(opal_pop_t *p, size_t n, const double *tau_abs, const double *tau_st,
                                     const double *birth_time);
/* chi of the first n slots (tests: a population that has not been pushed through a field yet) */
int opal_pop_upload_chi(opal_pop_t *p, size_t n, const double *chi);
/* the transverse coordinates y, z of the first n slots (electron.rs:18-19 / photon.rs:19-20), likewise */
int opal_pop_upload_yz(opal_pop_t *p, size_t n, const double *y, const double *z);
/* airy_ai_for_positive and the two scaled cross sections on the device for n inputs in host memory: k, p are
 * n x 4 (normalised four-momenta of photon and electron); out_* receive NaN where the reference returns None */
int opal_qed_airy_batch(size_t n, const double *x, double *out);
int opal_qed_cross_sections_batch(size_t n, const double *k, const double *p, const double *chi_gamma,
                                  const double *chi_e, double *out_absorption, double *out_stimulated);
/* photon_emission::sample / classical_sample, the emission random numbers and photon_emission::rate
 * evaluated on the device for n inputs in host memory (out: 3n / 4n / n doubles). */
int opal_qed_sample_batch(size_t n, const double *chi, const double *gamma, const double *r1, const double *r2,
                          const double *r3, int classical, double *out);
/* Electron::radiate's event (electron.rs:213-240) for n electrons with the three uniforms given: out[9 i ..] = the
 * photon's k[3], the electron's u[3] after the recoil, its gamma and chi, the formation length. */
int opal_qed_radiate_batch(size_t n, const double *chi, const double *ux, const double *uy, const double *uz, const double *r1,
                           const double *r2, const double *r3, double *out);
int opal_qed_randoms_batch(size_t n, uint64_t seed, const uint64_t *id, uint32_t step, double *out);
int opal_qed_rate_batch(size_t n, const double *chi, const double *gamma, double *out);

/* ------------------------------------------------------------------ Distribution functions
 * Population::write_data (mod.rs:383-568) and Histogram (hgram.rs:168-425).  `spec` is one entry of the
 * input file's `output:` list: "x", "x:px", "energy:(log;energy)", ...  The accessors, the min/max
 * pass and the weighted binning run on the GPU.  The reference's MPI all-reductions (hgram.rs:190-192,
 * 249, 260) are the caller's: range -> reduce min/max/count -> fill -> sum counts and totals -> finish
 * -> write.  Specs the reference silently skips (unknown quantity or weight) return OPAL_ERR_CONFIG. */
int opal_pop_hist_range(const opal_pop_t *p, const opal_grid_t *g, const char *spec, int *dim, double *min, double *max,
                        size_t *count);
int opal_pop_hist_fill(const opal_pop_t *p, const opal_grid_t *g, const char *spec, const double *gmin, const double *gmax,
                       size_t gnum, opal_hist_t **out);
size_t opal_hist_len(const opal_hist_t *h);
double *opal_hist_counts(opal_hist_t *h);          /* host array of opal_hist_len doubles (raw until finish) */
double opal_hist_total(const opal_hist_t *h);      /* total weight, hgram.rs:224 */
int opal_hist_set_total(opal_hist_t *h, double global_total);
int opal_hist_dims(const opal_hist_t *h, int *dim, size_t *nbins, double *min, double *bin_size);
int opal_hist_finish(opal_hist_t *h);              /* density normalisation, hgram.rs:255-259; once */
int opal_hist_write_fits(const opal_hist_t *h, const char *path);  /* hgram.rs:394-425 */
int opal_hist_free(opal_hist_t *h);
/* file name of mod.rs:527-559 (without CFITSIO's leading '!') */
int opal_hist_filename(const char *dir, size_t index, const char *pop_name, const char *spec, char *out, size_t len);
/* write_data for one subdomain: one FITS file per mappable spec */
int opal_pop_write_data(const opal_pop_t *p, const opal_grid_t *g, const char *dir, size_t index, const char *pop_name,
                        const char *const *specs, size_t nspecs);

/* ------------------------------------------------------------------------ Link
 * Neighbour exchange for one subdomain of `size`.  The library fills fixed-size
 * send buffers and calls `exchange` once per exchange point; the callee moves
 * send_left -> left neighbour's recv_right and send_right -> right neighbour's
 * recv_left (e.g. RCCL/NCCL or gloo send/recv) and returns 0.  Buffers are
 * device memory when device_buffers != 0 (stream-ordered transports such as
 * RCCL: the callee must order its transfers after/before work on `hip_stream`),
 * else pinned host memory (the library synchronises around the call). */
typedef int (*opal_exchange_fn)(void *user, const void *send_left, const void *send_right, void *recv_left,
                                void *recv_right, size_t nbytes);
int opal_link_create(opal_link_t **out, int rank, int size, int device_buffers, size_t particle_capacity,
                     opal_exchange_fn exchange, void *user, void *hip_stream);
/* The same with room for `species_slots` species (2: electrons, ions; 3: + photons) and
 * `fields_per_particle` doubles per particle (6: x ux uy uz w prev_x; 12: + chi work y z tau id, needed
 * when photon emission is on: the reference ships whole structs, mod.rs:308-354). */
int opal_link_create_layout(opal_link_t **out, int rank, int size, int device_buffers, size_t particle_capacity,
                            int species_slots, int fields_per_particle, opal_exchange_fn exchange, void *user,
                            void *hip_stream);
/* Native transport: the exchange is four RCCL point-to-point calls (send left, send right, receive right,
 * receive left) in one group on `hip_stream`, the stream the simulation runs on -- no host-language callback,
 * no second stream.  `librccl_path` is opened with dlopen (pass the copy the process already uses, e.g.
 * PyTorch's torch/lib/librccl.so); `unique_id` is the 128-byte ncclUniqueId that rank 0 obtained from
 * opal_rccl_unique_id and handed to every subdomain (any broadcast will do).  Collective: every subdomain
 * must call it. */
#define OPAL_RCCL_UNIQUE_ID_BYTES 128
int opal_rccl_unique_id(const char *librccl_path, void *id128);
int opal_link_create_rccl(opal_link_t **out, int rank, int size, size_t particle_capacity, int species_slots,
                          int fields_per_particle, const char *librccl_path, const void *unique_id, void *hip_stream);
/* Peer-mapped mailboxes: no transport is called at all -- the kernels of the step write their messages straight
 * into mailboxes in the neighbours' device memory (over xGMI between GPUs of one node) and wait for theirs
 * (peer_device.hpp).  Replaces the blocking MPI_Sendrecv pairs of yee.rs:358-409 and particle/mod.rs:303-354 with
 * zero launches and zero host calls per exchange on the fused step.  Set-up: every subdomain creates its link,
 * exports the 64-byte handle of its mailboxes (hipIpcGetMemHandle), the handles travel by any means (one
 * all-gather), and every subdomain connects with its left and right neighbour's handle; NULL for a neighbour means
 * "myself" (a lone periodic subdomain exchanging with itself).  When both neighbours are the same subdomain
 * (two subdomains, periodic) pass the same handle twice. */
#define OPAL_PEER_HANDLE_BYTES 64
int opal_link_create_peer(opal_link_t **out, int rank, int size, size_t particle_capacity, int species_slots,
                          int fields_per_particle, void *hip_stream);
int opal_link_peer_export(opal_link_t *l, void *handle64);
int opal_link_peer_connect(opal_link_t *l, const void *left_handle64, const void *right_handle64);
/* Synchronises the link's stream and reports: timeouts != 0 if a wait for a neighbour's message ever gave up (the
 * call then fails with OPAL_ERR_LINK), the kind of memory the mailboxes live in (0 uncached, 1 fine-grained, 2
 * plain), and the number of exchanges made so far. */
int opal_link_peer_status(opal_link_t *l, int *timeouts, int *alloc_kind, uint32_t *exchanges);
/* How long this subdomain has waited for its neighbours' messages inside the step's kernels (peer_device.hpp: peer_wait;
 * the blocking MPI_Recv of yee.rs:366-402 and mod.rs:311-347 in the reference): microseconds in total and the number of
 * waits, since creation or the last call with reset != 0.  On real hardware the wait is link latency where it is the same
 * few microseconds on every subdomain, imbalance or skew where some wait long and others not at all.  Drains the stream. */
int opal_link_peer_wait(opal_link_t *l, double *microseconds, uint64_t *waits, int reset);
/* n halo-only exchanges with the neighbours on the link's stream: 0 if they completed and -- links with device buffers
 * -- every halo word that arrived is the one the neighbour on that side sent in that round (each round's halo names its
 * sender, side and round: a late, stale or torn message fails); links with device buffers then make 4 n exchanges back
 * to back (the cadence of a run) and one more checked round.  Collective in the sense that the neighbours must make
 * the same calls. */
int opal_link_self_test(opal_link_t *l, int to_left, int to_right, int n);
/* The planning of message sizes and of the room populations reserve (link.cpp: link_plan) checked against its specification
 * on a link that exists in host memory only: 0, or an error naming the first exchange that was planned differently.  Touches
 * no device (the CPU-side tests call it). */
int opal_link_plan_selftest(void);
/* Ranks of the RCCL communicator behind a link made by opal_link_create_rccl (ncclCommCount): lets a caller
 * report that the subdomains really are joined by one RCCL communicator. */
int opal_link_rccl_ranks(const opal_link_t *l, int *out);
/* Per-edge sizes.  A message's body is laid out so that the leavers it carries sit in a prefix (link.cpp), and
 * every push counts the particles within 3 cells of each edge into the message header; from those counts --
 * which both ends of an edge hold, three exchanges later, without any extra traffic or host synchronisation --
 * the library sizes each edge's messages to what can physically cross it.  That needs a transport that takes
 * one size per edge: the left edge (send_left and recv_left) moves nbytes_left, the right edge nbytes_right.
 * Links made by opal_link_create_rccl have one; a callback transport opts in here.  Without it every
 * whole-message exchange ships all `particle_capacity` slots, as the plain opal_exchange_fn implies. */
typedef int (*opal_exchange2_fn)(void *user, const void *send_left, const void *send_right, void *recv_left,
                                 void *recv_right, size_t nbytes_left, size_t nbytes_right);
int opal_link_set_exchange2(opal_link_t *l, opal_exchange2_fn exchange2);
/* on = 0: ship all slots every time even though the transport could take sizes (diagnostic; default on) */
int opal_link_set_bounded(opal_link_t *l, int on);
/* Growth.  The reference's receive_vec (particle/mod.rs:308-354) takes messages of any size; a link has buffers.  With
 * growth on, an edge whose bound -- the particles within reach of the cut on both sides, the numbers both of its ends hold
 * (link.cpp: link_plan) -- exceeds its room gets larger buffers before the push that could overflow them is enqueued; on a
 * peer link the two ends hand each other the handle of larger mailboxes through a control line in the old ones, with no
 * host channel.  Default: on for peer links (what travels is only ever the leavers, so room costs memory, not time), off
 * for transports that are told a size (their messages follow the bound: a larger room means larger messages); a link whose
 * buffers belong to the caller (opal_link_set_buffers) cannot grow.  Without growth an overflow is the error it always was. */
int opal_link_set_growth(opal_link_t *l, int on);
/* room per species of the left / right edge now, and how often an edge of this link has grown */
int opal_link_growth(const opal_link_t *l, size_t *cap_left, size_t *cap_right, unsigned *grows);
/* bytes the last whole-message exchange moved over the left / right edge (0 where there is no neighbour) */
int opal_link_last_exchange_bytes(const opal_link_t *l, size_t *left, size_t *right);
int opal_link_free(opal_link_t *l);
/* bytes per direction of a message with every slot in use (what the buffers hold) and of its halo-only prefix */
size_t opal_link_message_bytes(const opal_link_t *l);
size_t opal_link_halo_bytes(const opal_link_t *l);
size_t opal_link_particle_capacity(const opal_link_t *l);
/* Replaces the four buffers the transport sees (send_left, send_right, recv_left, recv_right) by
 * caller-owned ones of >= opal_link_message_bytes() each -- device memory if device_buffers != 0,
 * host memory otherwise -- so that a transport can name them in its own terms (e.g. torch tensors). */
int opal_link_set_buffers(opal_link_t *l, void *send_left, void *send_right, void *recv_left, void *recv_right);
/* which: 0 send_left, 1 send_right, 2 recv_left, 3 recv_right */
void *opal_link_buffer(opal_link_t *l, int which);

/* ------------------------------------------------------------------ Simulation
 * main() of the reference (src/main.rs:48-286): reads the same YAML input file, builds grid +
 * populations for subdomain `rank` of `numtasks`, and runs the time loop (photon emission included,
 * absorption not). */
int opal_sim_create(opal_sim_t **out, const char *yaml_path, int rank, int numtasks, opal_link_t *link,
                    void *hip_stream);
int opal_sim_create_from_string(opal_sim_t **out, const char *yaml_text, const char *output_dir, int rank,
                                int numtasks, opal_link_t *link, void *hip_stream);
int opal_sim_free(opal_sim_t *s);
/* `n` iterations of main.rs:238-267 */
int opal_sim_step(opal_sim_t *s, size_t n);
/* the whole of main.rs:217-283, writing {i}_energy.dat / {i}_grid.dat / {i}_{species}_*.fits into the
 * output directory.  With more than one subdomain the output-time reductions of the reference
 * (MPI_Reduce of the energies, main.rs:23-31; gather of the grid rows to rank 0, yee.rs:749-779;
 * MPI_Allreduce of histogram ranges and counts, hgram.rs:190-192, 249, 260) go through the two
 * callbacks below, which every subdomain must have set (e.g. torch.distributed all_reduce / all_gather).
 *   allreduce: in place over all subdomains, op 0 = sum, 1 = min, 2 = max.
 *   gather: subdomain r contributes counts[r] doubles; `all` (rank 0 only, else may be null) receives
 *   them concatenated in rank order. */
typedef int (*opal_allreduce_fn)(void *user, double *values, size_t n, int op);
typedef int (*opal_gather_fn)(void *user, const double *local, size_t n_local, double *all, const size_t *counts,
                              int numtasks);
int opal_sim_set_collectives(opal_sim_t *s, opal_allreduce_fn allreduce, opal_gather_fn gather, void *user);
int opal_sim_run(opal_sim_t *s);
int opal_sim_sync(opal_sim_t *s);
double opal_sim_time(const opal_sim_t *s);
double opal_sim_dt(const opal_sim_t *s);
size_t opal_sim_total_steps(const opal_sim_t *s);      /* main.rs:186 */
size_t opal_sim_steps_between_outputs(const opal_sim_t *s); /* main.rs:187 */
size_t opal_sim_n_outputs(const opal_sim_t *s);
opal_grid_t *opal_sim_grid(opal_sim_t *s);
opal_pop_t *opal_sim_electrons(opal_sim_t *s);
opal_pop_t *opal_sim_ions(opal_sim_t *s);
opal_pop_t *opal_sim_photons(opal_sim_t *s); /* null unless qed.photon_emission is on */
/* write_energies, main.rs:23-42: em_field, electrons, ions (this subdomain's parts) */
int opal_sim_energies(opal_sim_t *s, double *em_field, double *electrons, double *ions);
/* 1 = use the fused advance+deposit pass (default), 0 = the reference's operator sequence */
int opal_sim_set_fused(opal_sim_t *s, int fused);
/* 1 = in the fused step of a decomposed run, push the particles near the subdomain's ends first, start the
 * neighbour exchange on a second stream and push the rest meanwhile; 0 (default) = push, then exchange.
 * In the one-GPU rehearsal (bench.py --self-exchange) the overlap costs more (a small extra launch, the RCCL
 * kernel competing with the push for CUs: 0.357 against 0.338 ms/step) than the 25 us exchange it hides;
 * it is there for links whose latency is larger than that. */
int opal_sim_set_overlap(opal_sim_t *s, int on);
size_t opal_sim_overlapped_steps(const opal_sim_t *s); /* how many steps so far ran that way (diagnostic) */
/* Cuts that follow the plasma (opt-in).  The reference partitions the domain once, at t = 0, by the electrons' initial
 * density (GridDesign::balanced, grid/mod.rs:157-206, called from main.rs:103-110); a laser that piles the plasma up or a
 * beam that crosses the domain then leaves most particles on one or two subdomains.  With every > 0, every `every`-th
 * step of opal_sim_step the subdomains all-reduce their particle loads (opal_sim_set_collectives' allreduce) and, when
 * max / mean exceeds `threshold` (>= 1), move the cuts by the reference's own partition rule applied to the particles
 * where they stand now -- each cut by `max_shift` cells at the most (0: 64), never out of the two subdomains it separates,
 * every subdomain keeping opal_grid_min_size() cells (yee.rs:783-785).  The cells that change hands travel with everything
 * in them: particles as leavers through the link (mod.rs:291-293, 366-368), grid columns through the all-reduce; the
 * subdomain sizes the grid rows are gathered by follow (yee.rs:749-779).  every = 0 (default): the reference's partition,
 * made once.  Results do not depend on where the cuts are (only the deposit's summation order does).
 * opal_sim_recut_now: one check + re-cut at once, whatever the threshold (collective; for callers that step by hand).
 * opal_sim_recut_stats: checks and re-cuts so far, max / mean at the last check and the worst since the previous call
 * (0: never checked), the subdomains' present sizes (numtasks entries), and the host time the step loop has spent in
 * checks and re-cuts so far, the draining of the device before a plan included (any pointer may be null). */
int opal_sim_set_recut(opal_sim_t *s, size_t every, double threshold, size_t max_shift);
int opal_sim_recut_now(opal_sim_t *s);
int opal_sim_recut_stats(opal_sim_t *s, size_t *checks, size_t *recuts, double *imbalance_last, double *imbalance_worst,
                         size_t *nx_of_every_subdomain, double *seconds);
/* Between two plans the cuts DRIFT: each interior cut carries a velocity learnt from what the plans keep asking of it (a beam
 * that crosses the domain at c, main.rs has no counterpart) and moves a cell at a time inside the step's own pushes, with no
 * plan, no all-reduce and no data from a neighbour.  cells_per_step_of_every_cut: numtasks + 1 entries (the outer ones 0);
 * drift_steps: steps in which some cut moved that way.  OPAL_RECUT_NO_DRIFT=1 in the environment keeps the cuts still
 * between plans. */
int opal_sim_drift_stats(opal_sim_t *s, double *cells_per_step_of_every_cut, size_t *drift_steps);
/* The plan of a re-cut, pure host arithmetic (no device): cuts[0 .. numtasks] are the first global cell of every
 * subdomain and one past the last; cost[c] the load of global cell c, most[c] the particle count of its fullest species
 * (null: cost); edge_room[i - 1] the particles per species cut i may move (null or negative: any number).  new_cuts gets
 * the partition of grid/mod.rs:170-176 on `cost`, held to the limits above; imbalance_* = max / mean of the subdomains'
 * cost before and after (may be null). */
int opal_recut_plan(int numtasks, const size_t *cuts, const double *cost, const double *most, size_t min_size,
                    size_t max_shift, const double *edge_room, size_t *new_cuts, double *imbalance_before,
                    double *imbalance_after);
/* A HIP stream of the library's own runtime, for hosts that have none to hand in (several subdomains inside one
 * process, one stream each: tests/_inproc_world.py).  Every `hip_stream` argument of this header takes it. */
int opal_stream_create(void **out);
int opal_stream_destroy(void *stream);
/* per-kernel timing with HIP events on the library's stream: stride 0 = off, n = time every n-th
 * launch of each tracked kernel (an odd n samples both species, which alternate) */
int opal_prof_enable(int stride);
int opal_prof_reset(void);
/* kernel ids: 0 = fused advance+deposit (one launch covers both species when the driver pairs them),
 * 1 = advance (charged species), 2 = deposit, 3 = field solve, 4 = re-binning, 5 = the photons' push (photon.rs:150-183),
 * 6 = the emission kernel (interactions.rs:45-107); launches = launches actually timed */
int opal_prof_get(int kernel, uint64_t *launches, double *total_ms);
/* launches of the kernel observed since the last reset while the timers were on (timed or not) */
int opal_prof_seen(int kernel, uint64_t *launches_seen);
/* launches of the fused push since the last reset (timers on), by the form of the kernel that ran: index = bit 0
 * electrostatic launch (uy, uz not streamed) | bit 1 rho deposited | bit 2 / bit 3 electrons' / ions' weight from the
 * launch arguments (not streamed) | bit 4 single-species launch; 32 entries.  Each form has its own algorithmic
 * bytes per particle-update, which is what a roofline fraction must be priced with. */
int opal_prof_forms(uint64_t *launches_by_form);

/* Text forms used by the reference's outputs: kind 0 = SimulationTime (setup.rs:421-438), 1 = Rust {:e}
 * (yee.rs:768), 2 = {:.6e} (main.rs:35-38), 3 = {} (yee.rs:768, last column) */
int opal_format(int kind, double v, char *buf, size_t len);

/* ------------------------------------------------------------------- Config
 * src/setup.rs: typed reads of `section.field` with the `constants` context. */
typedef struct opal_config opal_config_t;
int opal_config_from_file(opal_config_t **out, const char *path);
int opal_config_from_string(opal_config_t **out, const char *text);
int opal_config_free(opal_config_t *c);
int opal_config_contains(const opal_config_t *c, const char *section);
int opal_config_read_f64(const opal_config_t *c, const char *section, const char *field, double *out);
int opal_config_read_usize(const opal_config_t *c, const char *section, const char *field, size_t *out);
int opal_config_read_bool(const opal_config_t *c, const char *section, const char *field, int *out);
/* Config::func / func2 / func3 evaluated at a point (args beyond the arity are ignored) */
int opal_config_eval(const opal_config_t *c, const char *section, const char *field, int nargs,
                     const char *const *arg_names, const double *arg_values, double *out);
/* bare expression against the config's context (meval grammar) */
int opal_config_eval_expr(const opal_config_t *c, const char *expr, double *out);

#ifdef __cplusplus
}
#endif
#endif /* OPAL_AMD_H */
