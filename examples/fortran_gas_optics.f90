!> A Fortran caller of the library through the reference's own binding module (`use grtcode`,
!! fortran-bindings/grtcode_fortran.F90, compiled unchanged) and its C helper (malloc_structs.c, compiled
!! unchanged against include/): device, grid, gas optics with two molecules, Rayleigh scattering, add_optics,
!! and the optical properties copied back to Fortran arrays.
!!
!!   fortran_gas_optics HITRAN.par W0 WN DW [H2O_CTM_DIR O3_CTM_FILE]
!!
!! The column is analytic (9 levels) so that a test can rebuild it: p_k = 1 + 1012.25 (k/8)^2 mb,
!! T_k = 210 + 80 k/8 K, H2O = 5 (3000)^(k/8) ppmv, CO2 = 400 ppmv.
!! Output: per layer the sum over the grid of tau_gas, tau, omega*tau (17 significant digits).
program fortran_gas_optics
use, intrinsic :: iso_c_binding
use grtcode
implicit none

integer, parameter :: nlev = 9
type(Device_t) :: device
type(Grid_t) :: grid
type(MolecularLines_t) :: lbl
type(Optics_t) :: gas, ray, total
type(Optics_t), dimension(2) :: parts
real(kind=c_double) :: w0, wn, dw, w0_back, dw_back
integer(kind=c_int64_t) :: n
real(kind=c_double), dimension(nlev) :: p, t, x_h2o, x_co2
real(kind=c_double), dimension(:,:), allocatable :: tau_gas, tau, omega, g
character(len=1024) :: hitran, arg, h2o_dir, o3_file
integer :: k, nmol

call get_command_argument(1, hitran)
call get_command_argument(2, arg); read(arg, *) w0
call get_command_argument(3, arg); read(arg, *) wn
call get_command_argument(4, arg); read(arg, *) dw

do k = 1, nlev
  p(k) = 1.d0 + 1012.25d0*(real(k - 1, c_double)/real(nlev - 1, c_double))**2
  t(k) = 210.d0 + 80.d0*real(k - 1, c_double)/real(nlev - 1, c_double)
  x_h2o(k) = 5.d0*3000.d0**(real(k - 1, c_double)/real(nlev - 1, c_double))
  x_co2(k) = 400.d0
enddo

call check(create_device(device), "create_device")
call check(create_spectral_grid(grid, w0, wn, dw), "create_spectral_grid")
call check(spectral_grid_properties(grid, w0=w0_back, n=n, dw=dw_back), "spectral_grid_properties")
if (command_argument_count() .ge. 6) then
  call get_command_argument(5, h2o_dir)
  call get_command_argument(6, o3_file)
  call check(create_gas_optics(lbl, nlev, grid, device, trim(hitran), h2o_ctm_dir=trim(h2o_dir), &
                               o3_ctm_dir=trim(o3_file), optical_depth_method=2), "create_gas_optics")
else
  call check(create_gas_optics(lbl, nlev, grid, device, trim(hitran), optical_depth_method=2), "create_gas_optics")
endif
call check(add_molecule(lbl, H2O), "add_molecule H2O")
call check(add_molecule(lbl, CO2), "add_molecule CO2")
call check(num_molecules(lbl, nmol), "num_molecules")
call check(set_molecule_ppmv(lbl, H2O, x_h2o), "set_molecule_ppmv H2O")
call check(set_molecule_ppmv(lbl, CO2, x_co2), "set_molecule_ppmv CO2")
call check(create_optics(gas, nlev - 1, grid, device), "create_optics gas")
call check(create_optics(ray, nlev - 1, grid, device), "create_optics rayleigh")
call check(calculate_optics(lbl, p, t, gas), "calculate_optics")
call check(rayleigh_scattering(ray, p), "rayleigh_scattering")
parts(1) = gas
parts(2) = ray
total%optics = c_null_ptr
call check(malloc_struct(total%optics, 1), "malloc_struct")
call check(add_optics(parts, total), "add_optics")

allocate(tau_gas(n, nlev - 1), tau(n, nlev - 1), omega(n, nlev - 1), g(n, nlev - 1))
call check(optical_properties(gas, tau=tau_gas), "optical_properties gas")
call check(optical_properties(total, tau=tau, omega=omega, g=g), "optical_properties total")

write(*, "(a,i0,a,i0,a,es24.16,a,es24.16)") "grid: n=", n, " molecules=", nmol, " w0=", w0_back, " dw=", dw_back
do k = 1, nlev - 1
  write(*, "(a,i0,3(1x,es24.16))") "layer ", k, sum(tau_gas(:, k)), sum(tau(:, k)), sum(omega(:, k)*tau(:, k))
enddo

call check(destroy_optics(total), "destroy_optics total")
call check(destroy_optics(ray), "destroy_optics rayleigh")
call check(destroy_optics(gas), "destroy_optics gas")
call check(destroy_gas_optics(lbl), "destroy_gas_optics")
call check(destroy_spectral_grid(grid), "destroy_spectral_grid")

contains

subroutine check(code, what)
  integer(kind=c_int), intent(in) :: code
  character(len=*), intent(in) :: what
  character(kind=c_char, len=2048) :: buf
  integer(kind=c_int) :: rc
  if (code .ne. grtcode_success) then
    rc = grtcode_errstr(code, buf)
    write(*, "(a,a,a,i0)") "error in ", what, ": code ", code
    write(*, "(a)") trim(buf)
    stop 1
  endif
end subroutine check

end program fortran_gas_optics
